// rt_dp_replay: thin host driver that swaps the GPU evaluator in for simulated vectors.
//
//   rt_dp_replay <file.vec> [--dump out.vec] [--quiet] [--gpus N]
//
// Reads a vector file (format: raytracer_amd/vecfile.py), feeds the input records
// to a UnifiedDatapath instance the way the reference's testbench does
// (src/test/scala/raytracer_datapath/Datapath_test.scala:476-496: enqueueSeq on
// `in`, waitForValid / peek / step on `out`), and compares every output record
// bit for bit with the expected record in the file.  Exit status 0 = all equal.
// --dump writes the records that were computed, in the same format (e.g. to diff
// against vectors captured from a Verilator run of the RTL).
// --gpus N cuts the record stream into N parts where both accumulators are idle (rt_dp_shard_plan_records) and gives
// every part its own thread and its own UnifiedDatapath instance on GPU r % rt_dp_device_count(); the parts' output
// records, put together in order, are compared with the file like those of the single instance.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "unified_datapath.hpp"

using namespace raytracer_datapath;

static_assert(sizeof(EnhancedInputBundle) == 84 * 4, "input record is 84 words");
static_assert(sizeof(EnhancedOutputBundle) == 21 * 4, "output record is 21 words");

static const char *kOutNames[21] = {"opcode", "tmin_out[0]", "tmin_out[1]", "tmin_out[2]", "tmin_out[3]",
                                    "isIntersect[0]", "isIntersect[1]", "isIntersect[2]", "isIntersect[3]",
                                    "boxIndex[0]", "boxIndex[1]", "boxIndex[2]", "boxIndex[3]", "t_num", "t_denom",
                                    "triangle_hit", "euclidean_accumulator", "euclidean_reset_accum",
                                    "angular_dot_product", "angular_norm", "angular_reset_accum"};

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <file.vec> [--dump out.vec] [--quiet]\n", argv[0]);
        return 2;
    }
    std::string dump;
    bool quiet = false;
    int gpus = 1;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!strcmp(argv[i], "--quiet")) quiet = true;
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
    }
    if (gpus < 1 || gpus > 64) { fprintf(stderr, "--gpus must be 1..64\n"); return 2; }
    std::ifstream f(argv[1]);
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }

    std::string name = "stream", line;
    int extended = 1;
    std::vector<EnhancedInputBundle> ins;
    std::vector<EnhancedOutputBundle> want;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line.substr(1));
        if (line[0] == 'S') {
            std::string tok;
            ss >> name;
            while (ss >> tok)
                if (tok.rfind("extended=", 0) == 0) extended = atoi(tok.c_str() + 9);
        } else if (line[0] == 'I' || line[0] == 'O') {
            const size_t nw = line[0] == 'I' ? 84 : 21;
            std::vector<uint32_t> w;
            std::string tok;
            while (ss >> tok) w.push_back((uint32_t)strtoul(tok.c_str(), nullptr, 16));
            if (w.size() != nw) { fprintf(stderr, "%s: record with %zu words, expected %zu\n", argv[1], w.size(), nw); return 2; }
            if (line[0] == 'I') { EnhancedInputBundle b; memcpy(&b, w.data(), sizeof b); ins.push_back(b); }
            else { EnhancedOutputBundle b; memcpy(&b, w.data(), sizeof b); want.push_back(b); }
        } else {
            fprintf(stderr, "%s: unknown record tag '%c'\n", argv[1], line[0]);
            return 2;
        }
    }
    if (!want.empty() && want.size() != ins.size()) { fprintf(stderr, "%s: %zu inputs but %zu expected outputs\n", argv[1], ins.size(), want.size()); return 2; }

    size_t bad = 0;
    try {
        RaytracerParams p;
        if (extended) p.support_euclidean = 16;
        // one instance per part of the stream; with --gpus 1 this is the reference's single DUT
        std::vector<uint64_t> cuts(gpus + 1);
        if (rt_dp_shard_plan_records(ins.data(), ins.size(), (uint32_t)gpus, cuts.data()) != RT_DP_OK) throw std::runtime_error(rt_dp_last_error());
        const int devices = rt_dp_device_count() > 0 ? rt_dp_device_count() : 1;
        std::vector<std::vector<EnhancedOutputBundle>> part_out(gpus);
        std::vector<std::string> part_err(gpus);
        auto run_part = [&](int r) {
            try {
                UnifiedDatapath dut(p, gpus > 1 ? r % devices : -1);
                dut.in.enqueueSeq(std::vector<EnhancedInputBundle>(ins.begin() + cuts[r], ins.begin() + cuts[r + 1]));
                for (uint64_t i = cuts[r]; i < cuts[r + 1]; ++i) {
                    dut.out.waitForValid();
                    part_out[r].push_back(dut.out.bits());
                    dut.out.step();
                }
            } catch (const std::exception &e) {
                part_err[r] = e.what();
                if (part_err[r].empty()) part_err[r] = "error";
            }
        };
        if (gpus == 1) {
            run_part(0);
        } else {
            std::vector<std::thread> threads;
            for (int r = 0; r < gpus; ++r) threads.emplace_back(run_part, r);
            for (auto &t : threads) t.join();
        }
        std::vector<EnhancedOutputBundle> computed;
        for (int r = 0; r < gpus; ++r) {
            if (!part_err[r].empty()) throw std::runtime_error("part " + std::to_string(r) + ": " + part_err[r]);
            computed.insert(computed.end(), part_out[r].begin(), part_out[r].end());
        }
        if (gpus > 1 && !quiet) {
            printf("parts:");
            for (int r = 0; r < gpus; ++r) printf(" [%llu, %llu) on GPU %d", (unsigned long long)cuts[r], (unsigned long long)cuts[r + 1], r % devices);
            printf("\n");
        }
        std::ofstream out;
        if (!dump.empty()) {
            out.open(dump);
            out << "# computed by rt_dp_replay (" << rt_dp_version() << ")\nS " << name << " extended=" << extended << "\n";
        }
        for (size_t i = 0; i < ins.size(); ++i) {
            const EnhancedOutputBundle got = computed[i];
            uint32_t g[21], w[21];
            memcpy(g, &got, sizeof g);
            if (out.is_open()) {
                uint32_t iw[84];
                memcpy(iw, &ins[i], sizeof iw);
                out << "I";
                for (uint32_t x : iw) out << " " << std::hex << x;
                out << "\nO";
                for (uint32_t x : g) out << " " << std::hex << x;
                out << "\n";
            }
            if (want.empty()) continue;
            memcpy(w, &want[i], sizeof w);
            if (memcmp(g, w, sizeof g) != 0) {
                if (++bad <= 10 && !quiet)
                    for (int k = 0; k < 21; ++k)
                        if (g[k] != w[k]) printf("record %zu (opcode %u) %s: got %08x expected %08x\n", i, ins[i].opcode, kOutNames[k], g[k], w[k]);
            }
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "rt_dp_replay: %s\n", e.what());
        return 3;
    }
    if (!quiet) printf("%s: %zu records, %zu mismatching\n", name.c_str(), ins.size(), bad);
    return bad ? 1 : 0;
}
