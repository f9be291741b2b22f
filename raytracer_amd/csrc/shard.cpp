// Shard planner of the C ABI (include/rt_dp.h: rt_dp_shard_plan, rt_dp_shard_plan_records): where an opcode-sorted
// batch, or a record stream in arrival order, may be cut so that every part can be evaluated by its own context
// (its own GPU) from reset accumulators and the parts' results, put together in order, are the whole stream's.
//
// Box and triangle ops carry no state.  A distance beat carries the accumulator register(s) to the next beat of
// its opcode unless its reset bit is set (reference src/main/scala/raytracer_datapath/UnifiedDatapath.scala:
// 1133-1154 Euclidean, :1028-1064 angular), so a cut may only fall where the last beat of each distance opcode in
// front of it carried the reset bit: there the registers hold +0, which is what a fresh context starts from.
// Host arrays only, no GPU involved.
#include "../../include/rt_dp.h"

#include <cstdarg>
#include <cstdio>

namespace rtdp {
int api_fail(int code, const char *fmt, ...);  // api.cpp: sets rt_dp_last_error() of the calling thread
}

namespace {

constexpr uint32_t kResetBit = 1u << 16;
constexpr uint64_t kMaxUnits = 1ull << 40;

inline uint64_t even_cut(uint64_t n, uint32_t r, uint32_t world) {
    return (uint64_t)(((unsigned __int128)n * r) / world);
}

// Smallest pair boundary >= pos: an index i with i == 0, i == n, or reset(ctl[i - 1]).
uint64_t snap_forward(const uint32_t *ctl, uint64_t n, uint64_t pos) {
    if (pos == 0) return 0;
    for (uint64_t i = pos; i < n; ++i)
        if (ctl[i - 1] & kResetBit) return i;
    return n;
}

}  // namespace

extern "C" {

int rt_dp_shard_plan(const uint64_t n[4], const uint32_t *euc_ctl, const uint32_t *ang_ctl, uint32_t world, uint32_t flags,
                     rt_dp_shard_range *out) {
    if (!n || !out) return rtdp::api_fail(RT_DP_ENULL, "rt_dp_shard_plan: NULL argument");
    if (world == 0 || world > 65536u) return rtdp::api_fail(RT_DP_EINVAL, "rt_dp_shard_plan: world must be 1..65536");
    if (flags & ~(RT_DP_F_BOX_TILED | RT_DP_F_TRI_TILED)) return rtdp::api_fail(RT_DP_EINVAL, "rt_dp_shard_plan: unknown flag bits 0x%x", flags);
    for (int op = 0; op < 4; ++op)
        if (n[op] > kMaxUnits) return rtdp::api_fail(RT_DP_EINVAL, "rt_dp_shard_plan: n[%d] is not representable (limit 2^40)", op);
    if (n[RT_DP_OP_EUCLIDEAN] && !euc_ctl) return rtdp::api_fail(RT_DP_ENULL, "rt_dp_shard_plan: euc_ctl is NULL for a non-empty segment");
    if (n[RT_DP_OP_ANGULAR] && !ang_ctl) return rtdp::api_fail(RT_DP_ENULL, "rt_dp_shard_plan: ang_ctl is NULL for a non-empty segment");

    for (int op = 0; op < 4; ++op) {
        const bool beats = (op == RT_DP_OP_EUCLIDEAN || op == RT_DP_OP_ANGULAR);
        const bool tiled = (op == RT_DP_OP_QUADBOX && (flags & RT_DP_F_BOX_TILED)) || (op == RT_DP_OP_TRIANGLE && (flags & RT_DP_F_TRI_TILED));
        const uint32_t *ctl = (op == RT_DP_OP_EUCLIDEAN) ? euc_ctl : ang_ctl;
        const uint64_t tiles = (n[op] + RT_DP_TILE_OPS - 1) / RT_DP_TILE_OPS;
        uint64_t prev = 0;
        for (uint32_t r = 0; r < world; ++r) {
            uint64_t stop;
            if (r + 1 == world) {
                stop = n[op];
            } else if (tiled) {  // whole tiles: a rank's ops are then one contiguous slice of the tiled buffer
                stop = even_cut(tiles, r + 1, world) * RT_DP_TILE_OPS;
                if (stop > n[op]) stop = n[op];
            } else {
                stop = even_cut(n[op], r + 1, world);
                if (beats) stop = snap_forward(ctl, n[op], stop);
            }
            if (stop < prev) stop = prev;
            out[r].start[op] = prev;
            out[r].stop[op] = stop;
            prev = stop;
        }
    }
    return RT_DP_OK;
}

int rt_dp_shard_plan_records(const rt_dp_input_bundle *records, uint64_t n, uint32_t world, uint64_t *cuts) {
    if (!cuts || (n && !records)) return rtdp::api_fail(RT_DP_ENULL, "rt_dp_shard_plan_records: NULL argument");
    if (world == 0 || world > 65536u) return rtdp::api_fail(RT_DP_EINVAL, "rt_dp_shard_plan_records: world must be 1..65536");
    if (n > kMaxUnits) return rtdp::api_fail(RT_DP_EINVAL, "rt_dp_shard_plan_records: n is not representable (limit 2^40)");
    cuts[0] = 0;
    uint32_t r = 1;
    bool open_e = false, open_a = false;  // a vector pair of that opcode is under way at position i
    for (uint64_t i = 0; i <= n && r < world; ++i) {
        while (r < world && !open_e && !open_a && i >= even_cut(n, r, world)) cuts[r++] = i;
        if (i == n) break;
        const rt_dp_input_bundle &b = records[i];
        if (b.opcode > 3u) return rtdp::api_fail(RT_DP_EOPCODE, "record %llu has opcode %u; the RTL's opcode field is 2 bits wide", (unsigned long long)i, b.opcode);
        if (b.opcode == RT_DP_OP_EUCLIDEAN) open_e = (b.euclidean_reset_accum & 1u) == 0u;
        else if (b.opcode == RT_DP_OP_ANGULAR) open_a = (b.euclidean_reset_accum & 1u) == 0u;
    }
    while (r <= world) cuts[r++] = n;  // no admissible cut left: the remaining parts are empty
    return RT_DP_OK;
}

}  // extern "C"
