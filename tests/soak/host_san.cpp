// Host-side exerciser for a librt_dp.so whose HOST code was built with AddressSanitizer and
// UndefinedBehaviorSanitizer (tests/soak/host_san.sh builds both and runs this on a GPU box).
// It walks the entry points that take host memory -- plain and tiled batches, record streams
// on both sides of the pinned-staging threshold and in every stream mode, the host builders,
// error paths, contexts created and destroyed, the per-thread default context -- with sizes
// around every internal boundary.  The sanitizers are the checker; on top of that the tiled
// results must equal the plain ones and every stream mode must give the same records.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include <algorithm>
#include <numeric>

#include <hip/hip_runtime_api.h>

#include "rt_dp.h"

static int g_fail = 0;
#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                     \
            std::printf("\n");                            \
            ++g_fail;                                     \
        }                                                 \
    } while (0)

struct Rng {
    std::mt19937 g;
    explicit Rng(uint32_t seed) : g(seed) {}
    float f(float lo, float hi) { return std::uniform_real_distribution<float>(lo, hi)(g); }
    uint32_t u(uint32_t n) { return std::uniform_int_distribution<uint32_t>(0, n - 1)(g); }
};

struct HostBatch {
    uint64_t n_box, n_tri, n_euc, n_ang;
    std::vector<std::vector<float>> box;   // 34 operand arrays
    std::vector<std::vector<float>> tri;   // 15 float arrays: org, shear, vtx
    std::vector<uint32_t> k;
    std::vector<float> ea, eb, aa, ab;
    std::vector<uint32_t> ectl, actl;
};

static HostBatch make(uint64_t n_box, uint64_t n_tri, uint64_t n_euc, uint64_t n_ang, uint32_t seed) {
    Rng r(seed);
    HostBatch h{n_box, n_tri, n_euc, n_ang, {}, {}, {}, {}, {}, {}, {}, {}, {}};
    h.box.assign(34, std::vector<float>(n_box));
    for (auto &a : h.box) for (auto &x : a) x = r.f(-4.0f, 4.0f);
    for (uint64_t i = 0; i < n_box; ++i) {
        for (int d = 0; d < 3; ++d) h.box[6 + d][i] = 1.0f / h.box[3 + d][i];
        h.box[9][i] = 4000.0f;
        for (int b = 0; b < 4; ++b) for (int d = 0; d < 3; ++d) h.box[10 + 6 * b + 3 + d][i] = h.box[10 + 6 * b + d][i] + r.f(0.0f, 3.0f);
    }
    h.tri.assign(15, std::vector<float>(n_tri));
    for (auto &a : h.tri) for (auto &x : a) x = r.f(-2.0f, 2.0f);
    h.k.resize(n_tri);
    for (auto &x : h.k) { const uint32_t kz = r.u(3); x = ((kz + 1) % 3) | (((kz + 2) % 3) << 2) | (kz << 4); }
    h.ea.resize(n_euc * 16); h.eb.resize(n_euc * 16); h.ectl.resize(n_euc);
    for (auto &x : h.ea) x = r.f(-1.0f, 1.0f);
    for (auto &x : h.eb) x = r.f(-1.0f, 1.0f);
    for (auto &x : h.ectl) x = RT_DP_CTL(0xFFFFu, r.u(5) == 0);
    h.aa.resize(n_ang * 8); h.ab.resize(n_ang * 8); h.actl.resize(n_ang);
    for (auto &x : h.aa) x = r.f(-1.0f, 1.0f);
    for (auto &x : h.ab) x = r.f(-1.0f, 1.0f);
    for (auto &x : h.actl) x = RT_DP_CTL(0xFFu, r.u(7) == 0);
    return h;
}

struct HostResults {
    std::vector<float> tmin[4], t_num, t_den, eacc, adot, anorm;
    std::vector<uint32_t> flags, hit;
};

static void fill_plain(const HostBatch &h, rt_dp_op_batch &in) {
    std::memset(&in, 0, sizeof in);
    in.struct_size = sizeof in;
    in.n[RT_DP_OP_QUADBOX] = h.n_box; in.n[RT_DP_OP_TRIANGLE] = h.n_tri; in.n[RT_DP_OP_EUCLIDEAN] = h.n_euc; in.n[RT_DP_OP_ANGULAR] = h.n_ang;
    const float **bp = reinterpret_cast<const float **>(&in.box);
    for (int f = 0; f < 34; ++f) bp[f] = h.box[f].data();
    for (int d = 0; d < 3; ++d) { in.tri.org[d] = h.tri[d].data(); in.tri.shear[d] = h.tri[3 + d].data(); }
    for (int v = 0; v < 3; ++v) for (int d = 0; d < 3; ++d) in.tri.vtx[v][d] = h.tri[6 + 3 * v + d].data();
    in.tri.k = h.k.data();
    in.euc.a = h.ea.data(); in.euc.b = h.eb.data(); in.euc.ctl = h.ectl.data();
    in.ang.a = h.aa.data(); in.ang.b = h.ab.data(); in.ang.ctl = h.actl.data();
}

static void fill_results(const HostBatch &h, HostResults &r, rt_dp_result_batch &out) {
    std::memset(&out, 0, sizeof out);
    out.struct_size = sizeof out;
    for (int k = 0; k < 4; ++k) { r.tmin[k].assign(h.n_box, -1.0f); out.box.tmin_out[k] = r.tmin[k].data(); }
    r.flags.assign(h.n_box, 0xDEADu); out.box.flags = r.flags.data();
    r.t_num.assign(h.n_tri, -1.0f); r.t_den.assign(h.n_tri, -1.0f); r.hit.assign(h.n_tri, 7u);
    out.tri.t_num = r.t_num.data(); out.tri.t_denom = r.t_den.data(); out.tri.hit = r.hit.data();
    r.eacc.assign(h.n_euc, -1.0f); out.euc.accumulator = r.eacc.data();
    r.adot.assign(h.n_ang, -1.0f); r.anorm.assign(h.n_ang, -1.0f);
    out.ang.dot_product = r.adot.data(); out.ang.norm = r.anorm.data();
}

static bool same_bits(const float *a, const float *b, size_t n) { return n == 0 || std::memcmp(a, b, n * 4) == 0; }

static void plain_and_tiled(rt_dp_ctx *ctx, uint64_t n, uint32_t seed) {
    const HostBatch h = make(n, n + 3, n / 2 + 1, n / 3, seed);
    rt_dp_op_batch in;
    rt_dp_result_batch out;
    HostResults plain;
    fill_plain(h, in);
    fill_results(h, plain, out);
    float acc0[3] = {0.5f, -0.25f, 2.0f};
    CHECK(rt_dp_set_accum(ctx, acc0) == RT_DP_OK, "set_accum: %s", rt_dp_last_error());
    CHECK(rt_dp_eval_ctx(ctx, &in, &out) == RT_DP_OK, "plain eval n=%llu: %s", (unsigned long long)n, rt_dp_last_error());
    float acc1[3];
    CHECK(rt_dp_get_accum(ctx, acc1) == RT_DP_OK, "get_accum");

    // the same ops with the box and triangle segments in tiled form, host buffers
    const uint64_t nb = h.n_box, nt = h.n_tri;
    std::vector<float> box_in(RT_DP_TILED_WORDS(nb, RT_DP_BOX_TILE_IN_WORDS), 777.0f), tri_in(RT_DP_TILED_WORDS(nt, RT_DP_TRI_TILE_IN_WORDS), 777.0f);
    std::vector<uint32_t> box_out(RT_DP_TILED_WORDS(nb, RT_DP_BOX_TILE_OUT_WORDS), 0xABABABABu), tri_out(RT_DP_TILED_WORDS(nt, RT_DP_TRI_TILE_OUT_WORDS), 0xABABABABu);
    for (uint64_t i = 0; i < nb; ++i)
        for (uint32_t f = 0; f < 34; ++f) box_in[RT_DP_TILED_INDEX(i, f, 34u)] = h.box[f][i];
    for (uint64_t i = 0; i < nt; ++i) {
        for (uint32_t f = 0; f < 15; ++f) tri_in[RT_DP_TILED_INDEX(i, f, 16u)] = h.tri[f][i];
        std::memcpy(&tri_in[RT_DP_TILED_INDEX(i, 15u, 16u)], &h.k[i], 4);
    }
    HostResults tiled;
    fill_results(h, tiled, out);
    in.flags = RT_DP_F_BOX_TILED | RT_DP_F_TRI_TILED;
    in.box.org[0] = box_in.data();
    in.tri.org[0] = tri_in.data();
    out.box.tmin_out[0] = reinterpret_cast<float *>(box_out.data());
    out.tri.t_num = reinterpret_cast<float *>(tri_out.data());
    CHECK(rt_dp_set_accum(ctx, acc0) == RT_DP_OK, "set_accum");
    CHECK(rt_dp_eval_ctx(ctx, &in, &out) == RT_DP_OK, "tiled eval n=%llu: %s", (unsigned long long)n, rt_dp_last_error());
    uint64_t bad = 0;
    for (uint64_t i = 0; i < nb; ++i) {
        for (uint32_t k = 0; k < 4; ++k) {
            uint32_t want;
            std::memcpy(&want, &plain.tmin[k][i], 4);
            bad += box_out[RT_DP_TILED_INDEX(i, k, 5u)] != want;
        }
        bad += box_out[RT_DP_TILED_INDEX(i, 4u, 5u)] != plain.flags[i];
    }
    for (uint64_t i = 0; i < nt; ++i) {
        uint32_t a, b;
        std::memcpy(&a, &plain.t_num[i], 4);
        std::memcpy(&b, &plain.t_den[i], 4);
        bad += tri_out[RT_DP_TILED_INDEX(i, 0u, 3u)] != a;
        bad += tri_out[RT_DP_TILED_INDEX(i, 1u, 3u)] != b;
        bad += tri_out[RT_DP_TILED_INDEX(i, 2u, 3u)] != plain.hit[i];
    }
    // what lies behind the last op of the last tile is the caller's
    for (uint64_t i = nb; i < (nb + RT_DP_TILE_OPS - 1) / RT_DP_TILE_OPS * RT_DP_TILE_OPS; ++i)
        for (uint32_t k = 0; k < 5; ++k) bad += box_out[RT_DP_TILED_INDEX(i, k, 5u)] != 0xABABABABu;
    for (uint64_t i = nt; i < (nt + RT_DP_TILE_OPS - 1) / RT_DP_TILE_OPS * RT_DP_TILE_OPS; ++i)
        for (uint32_t k = 0; k < 3; ++k) bad += tri_out[RT_DP_TILED_INDEX(i, k, 3u)] != 0xABABABABu;
    bad += !same_bits(plain.eacc.data(), tiled.eacc.data(), h.n_euc);
    bad += !same_bits(plain.adot.data(), tiled.adot.data(), h.n_ang);
    bad += !same_bits(plain.anorm.data(), tiled.anorm.data(), h.n_ang);
    float acc2[3];
    CHECK(rt_dp_get_accum(ctx, acc2) == RT_DP_OK, "get_accum");
    bad += !same_bits(acc1, acc2, 3);
    CHECK(bad == 0, "tiled != plain at n=%llu (%llu differences)", (unsigned long long)n, (unsigned long long)bad);
}

static std::vector<rt_dp_input_bundle> records(uint64_t n, uint32_t seed) {
    Rng r(seed);
    std::vector<rt_dp_input_bundle> v(n);
    for (auto &b : v) {
        uint32_t *w = reinterpret_cast<uint32_t *>(&b);
        for (size_t k = 0; k < sizeof b / 4; ++k) {
            const float x = r.f(-3.0f, 3.0f);
            std::memcpy(&w[k], &x, 4);
        }
        b.opcode = r.u(4);
        b.ray.kz = r.u(3); b.ray.kx = (b.ray.kz + 1) % 3; b.ray.ky = (b.ray.kz + 2) % 3;
        b.euclidean_mask = 0xFFFFu;
        b.euclidean_reset_accum = r.u(4) == 0;
    }
    return v;
}

static void streams(rt_dp_ctx *ctx) {
    // 8 MiB of pinned staging = 19972 records in + out: sizes on both sides of it
    for (uint64_t n : {0ull, 1ull, 63ull, 64ull, 65ull, 97ull, 2049ull, 19900ull, 19973ull, 40000ull}) {
        const auto in = records(n, 1000u + (uint32_t)n);
        std::vector<rt_dp_output_bundle> ref;
        for (int mode : {1, 2, 3, 0}) {
            CHECK(rt_dp_set_option(ctx, "stream_mode", mode) == RT_DP_OK, "stream_mode %d", mode);
            CHECK(rt_dp_reset(ctx) == RT_DP_OK, "reset");
            std::vector<rt_dp_output_bundle> out(n);
            if (n) std::memset(out.data(), 0x5A, n * sizeof(rt_dp_output_bundle));
            const int rc = rt_dp_eval_bundles(ctx, in.data(), out.data(), n);
            CHECK(rc == RT_DP_OK, "eval_bundles n=%llu mode %d: %s", (unsigned long long)n, mode, rt_dp_last_error());
            if (mode == 1) ref = out;
            else CHECK(n == 0 || std::memcmp(ref.data(), out.data(), n * sizeof(rt_dp_output_bundle)) == 0, "mode %d != mode 1 at n=%llu", mode, (unsigned long long)n);
        }
    }
    // a stream in two pieces carries the registers across the calls
    const auto in = records(5000, 77);
    std::vector<rt_dp_output_bundle> whole(5000), parts(5000);
    CHECK(rt_dp_reset(ctx) == RT_DP_OK, "reset");
    CHECK(rt_dp_eval_bundles(ctx, in.data(), whole.data(), 5000) == RT_DP_OK, "whole");
    CHECK(rt_dp_reset(ctx) == RT_DP_OK, "reset");
    CHECK(rt_dp_eval_bundles(ctx, in.data(), parts.data(), 1234) == RT_DP_OK, "part 1");
    CHECK(rt_dp_eval_bundles(ctx, in.data() + 1234, parts.data() + 1234, 5000 - 1234) == RT_DP_OK, "part 2");
    CHECK(std::memcmp(whole.data(), parts.data(), 5000 * sizeof(rt_dp_output_bundle)) == 0, "two pieces != one stream");
    // an opcode outside the RTL's 2-bit field is refused
    auto bad = records(100, 5);
    bad[57].opcode = 9;
    std::vector<rt_dp_output_bundle> out(100);
    CHECK(rt_dp_eval_bundles(ctx, bad.data(), out.data(), 100) == RT_DP_EOPCODE, "opcode 9 accepted");
}

static void errors(rt_dp_ctx *ctx) {
    const HostBatch h = make(10, 10, 10, 10, 3);
    rt_dp_op_batch in;
    rt_dp_result_batch out;
    HostResults r;
    fill_plain(h, in);
    fill_results(h, r, out);
    CHECK(rt_dp_eval_ctx(ctx, nullptr, &out) != RT_DP_OK, "NULL batch accepted");
    CHECK(rt_dp_eval_ctx(ctx, &in, nullptr) != RT_DP_OK, "NULL results accepted");
    in.struct_size = 8;
    CHECK(rt_dp_eval_ctx(ctx, &in, &out) != RT_DP_OK, "short struct accepted");
    in.struct_size = sizeof in;
    in.flags = 0x100u;
    CHECK(rt_dp_eval_ctx(ctx, &in, &out) != RT_DP_OK, "unknown flag accepted");
    in.flags = 0;
    in.box.extent = nullptr;
    CHECK(rt_dp_eval_ctx(ctx, &in, &out) != RT_DP_OK, "NULL operand accepted");
    in.box.extent = reinterpret_cast<const float *>(reinterpret_cast<const char *>(h.box[9].data()) + 1);
    CHECK(rt_dp_eval_ctx(ctx, &in, &out) == RT_DP_EALIGN, "misaligned operand accepted");
    in.box.extent = h.box[9].data();
    CHECK(rt_dp_eval_ctx(ctx, &in, &out) == RT_DP_OK, "the repaired batch: %s", rt_dp_last_error());
    CHECK(rt_dp_set_option(ctx, "no_such_knob", 1) != RT_DP_OK, "unknown knob accepted");
    CHECK(rt_dp_set_option(ctx, "variant", 1234) != RT_DP_OK, "unknown variant accepted");
    int v = -1;
    CHECK(rt_dp_get_option(ctx, "box_wgs_per_cu", &v) == RT_DP_OK && v >= 0, "box_wgs_per_cu");
    CHECK(rt_dp_get_option(ctx, "vec_chain_giveups", &v) == RT_DP_OK && v == 0, "vec_chain_giveups = %d", v);
    CHECK(rt_dp_get_option(ctx, "stream_giveups", &v) == RT_DP_OK && v == 0, "stream_giveups = %d", v);
    rt_dp_config cfg{sizeof(rt_dp_config), -1, 1, 1, 16, 0};
    rt_dp_ctx *c2 = nullptr;
    CHECK(rt_dp_create(&cfg, &c2) != RT_DP_OK && c2 == nullptr, "io_recorded_float = 1 accepted");
    cfg.io_recorded_float = 0; cfg.support_euclidean = 8;
    CHECK(rt_dp_create(&cfg, &c2) != RT_DP_OK && c2 == nullptr, "support_euclidean = 8 accepted");
    cfg.support_euclidean = 0;
    CHECK(rt_dp_create(&cfg, &c2) == RT_DP_OK && c2 != nullptr, "baseline datapath: %s", rt_dp_last_error());
    if (c2) {
        fill_results(h, r, out);
        CHECK(rt_dp_eval_ctx(c2, &in, &out) == RT_DP_OK, "baseline eval: %s", rt_dp_last_error());
        bool zeros = true;
        for (float x : r.eacc) zeros &= (x == 0.0f);
        for (float x : r.adot) zeros &= (x == 0.0f);
        CHECK(zeros, "baseline datapath wrote distance results");
        rt_dp_destroy(c2);
    }
}

static void builders() {
    Rng r(11);
    for (uint32_t n : {1u, 2u, 127u, 128u, 129u, 5000u}) {
        std::vector<float> pts((size_t)n * 16);
        for (auto &x : pts) x = r.f(-1.0f, 1.0f);
        const uint32_t leaf = 128, nl = rt_dp_knn_tree_leaves(n, leaf);
        std::vector<float> op((size_t)nl * leaf * 16), lo((size_t)nl * 16), hi((size_t)nl * 16), sv(nl > 1 ? nl - 1 : 1);
        std::vector<uint32_t> idx((size_t)nl * leaf), sd(nl > 1 ? nl - 1 : 1), home(64);
        CHECK(rt_dp_knn_tree_build_host(pts.data(), n, leaf, 100, op.data(), idx.data(), lo.data(), hi.data(), sd.data(), sv.data()) == RT_DP_OK,
              "knn_tree_build_host n=%u: %s", n, rt_dp_last_error());
        uint32_t live = 0;
        for (uint32_t x : idx) live += x != 0xFFFFFFFFu;
        CHECK(live == n, "k-d index holds %u of %u points", live, n);
        std::vector<float> q(64 * 16);
        for (auto &x : q) x = r.f(-1.0f, 1.0f);
        CHECK(rt_dp_knn_home_leaf_host(q.data(), 64, nl, sd.data(), sv.data(), home.data()) == RT_DP_OK, "home_leaf_host");
        for (uint32_t x : home) CHECK(x < nl, "home leaf %u of %u", x, nl);
    }
    for (uint32_t n : {1u, 3u, 4u, 5u, 17u, 4097u}) {
        std::vector<float> tris((size_t)n * 9);
        for (auto &x : tris) x = r.f(-5.0f, 5.0f);
        const uint32_t rows = rt_dp_bvh4_max_nodes(n);
        std::vector<float> nodes((size_t)rows * 32), ot((size_t)n * 12);
        std::vector<uint32_t> order(n);
        uint32_t used = 0;
        CHECK(rt_dp_bvh4_build_host(tris.data(), n, 4, nodes.data(), &used, ot.data(), order.data()) == RT_DP_OK, "bvh4_build_host n=%u: %s", n, rt_dp_last_error());
        CHECK(used >= 1 && used <= rows, "bvh rows %u of %u", used, rows);
        std::vector<char> seen(n, 0);
        for (uint32_t x : order) { CHECK(x < n && !seen[x], "bvh order is not a permutation"); if (x < n) seen[x] = 1; }
    }
    const uint64_t po[5] = {0, 0, 16, 17, 50};
    uint64_t bo[5];
    CHECK(rt_dp_beat_offsets_host(po, 4, 16, bo) == 5 && bo[0] == 0 && bo[1] == 0 && bo[2] == 1 && bo[3] == 2 && bo[4] == 5, "beat_offsets_host");
}


// ---- entry points that take device memory: their host halves (argument checks, scratch growth, the leaf-major
// search's host-side rounds, the traversal's queues) run under the sanitizers too
template <typename T> struct Dev {
    T *p = nullptr;
    size_t n = 0;
    explicit Dev(size_t count) : n(count) { if (hipMalloc(reinterpret_cast<void **>(&p), (count ? count : 1) * sizeof(T)) != hipSuccess) { std::printf("hipMalloc failed\n"); std::exit(3); } }
    explicit Dev(const std::vector<T> &h) : Dev(h.size()) { up(h); }
    ~Dev() { (void)hipFree(p); }
    void up(const std::vector<T> &h) { if (!h.empty() && hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) std::exit(3); }
    std::vector<T> down() const { std::vector<T> h(n); if (n && hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) std::exit(3); return h; }
    Dev(const Dev &) = delete;
    Dev &operator=(const Dev &) = delete;
};

static void device_knn(rt_dp_ctx *ctx) {
    Rng r(21);
    for (uint32_t n : {300u, 20000u}) {
        const uint32_t nq = 700, leaf = 128, nl = rt_dp_knn_tree_leaves(n, leaf);
        std::vector<float> pts((size_t)n * 16), q((size_t)nq * 16);
        for (auto &x : pts) x = r.f(-1.0f, 1.0f);
        for (auto &x : q) x = r.f(-1.0f, 1.0f);
        std::vector<float> tp((size_t)nl * leaf * 16), lo((size_t)nl * 16), hi((size_t)nl * 16), sv(nl > 1 ? nl - 1 : 1);
        std::vector<uint32_t> idx((size_t)nl * leaf), sd(nl > 1 ? nl - 1 : 1), home(nq);
        CHECK(rt_dp_knn_tree_build_host(pts.data(), n, leaf, 0, tp.data(), idx.data(), lo.data(), hi.data(), sd.data(), sv.data()) == RT_DP_OK, "tree build");
        CHECK(rt_dp_knn_home_leaf_host(q.data(), nq, nl, sd.data(), sv.data(), home.data()) == RT_DP_OK, "home leaf");
        std::vector<uint32_t> order(nq);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return home[a] < home[b]; });
        std::vector<float> qs((size_t)nq * 16);
        std::vector<uint32_t> hs(nq);
        for (uint32_t i = 0; i < nq; ++i) { std::memcpy(&qs[(size_t)i * 16], &q[(size_t)order[i] * 16], 64); hs[i] = home[order[i]]; }
        Dev<float> d_pts(pts), d_q(qs), d_tp(tp), d_lo(lo), d_hi(hi), d_dist((size_t)nq * 16), d_dist2((size_t)nq * 16), d_dist3((size_t)nq * 16);
        Dev<uint32_t> d_idx(idx), d_home(hs), d_out((size_t)nq * 16), d_out2((size_t)nq * 16), d_out3((size_t)nq * 16), d_bound(nq);
        CHECK(rt_dp_knn_euclid16(ctx, d_pts.p, n, d_q.p, nq, 0, 1, d_dist.p, d_out.p, nullptr) == RT_DP_OK, "knn exhaustive: %s", rt_dp_last_error());
        rt_dp_knn_tree tree{sizeof(rt_dp_knn_tree), nl, leaf, 0, d_tp.p, d_idx.p, d_lo.p, d_hi.p};
        CHECK(rt_dp_knn_tree_euclid16(ctx, &tree, d_q.p, d_home.p, nq, 1, d_bound.p, d_dist2.p, d_out2.p, nullptr) == RT_DP_OK, "knn tree walk: %s", rt_dp_last_error());
        uint64_t stats[4] = {0, 0, 0, 0};
        CHECK(rt_dp_knn_tree_grouped_euclid16(ctx, &tree, d_q.p, d_home.p, nq, 8, d_dist3.p, d_out3.p, stats) == RT_DP_OK, "knn leaf-major: %s", rt_dp_last_error());
        CHECK(rt_dp_synchronize(ctx) == RT_DP_OK, "synchronize");
        const auto a = d_dist.down(), b = d_dist2.down(), c = d_dist3.down();
        const auto ia = d_out.down(), ib = d_out2.down(), ic = d_out3.down();
        CHECK(same_bits(a.data(), b.data(), a.size()) && ia == ib, "tree walk != exhaustive at n=%u", n);
        CHECK(same_bits(a.data(), c.data(), a.size()) && ia == ic, "leaf-major != exhaustive at n=%u", n);
        // the index built on the device, and the angular ranking
        Dev<float> b_tp((size_t)nl * leaf * 16), b_lo((size_t)nl * 16), b_hi((size_t)nl * 16), b_sv(nl > 1 ? nl - 1 : 1), d_neg((size_t)nq * 16);
        Dev<uint32_t> b_idx((size_t)nl * leaf), b_sd(nl > 1 ? nl - 1 : 1), b_home(nq);
        CHECK(rt_dp_knn_tree_build_device(ctx, d_pts.p, n, leaf, 0, b_tp.p, b_idx.p, b_lo.p, b_hi.p, b_sd.p, b_sv.p) == RT_DP_OK, "tree build on the device: %s", rt_dp_last_error());
        CHECK(rt_dp_knn_home_leaf(ctx, d_q.p, nq, nl, b_sd.p, b_sv.p, b_home.p, nullptr) == RT_DP_OK, "home leaf on the device: %s", rt_dp_last_error());
        CHECK(rt_dp_knn_angular16(ctx, d_pts.p, n, d_q.p, nq, 0, 1, d_neg.p, d_out2.p, nullptr) == RT_DP_OK, "knn angular: %s", rt_dp_last_error());
        CHECK(rt_dp_synchronize(ctx) == RT_DP_OK, "synchronize");
        // the angular ranking through the tree: the exhaustive lists again
        Dev<float> d_neg2((size_t)nq * 16);
        Dev<uint32_t> d_bound2(nq), d_out4((size_t)nq * 16);
        Dev<unsigned long long> d_scanned(1);
        CHECK(hipMemset(d_scanned.p, 0, 8) == hipSuccess, "memset");
        CHECK(rt_dp_knn_tree_angular16(ctx, &tree, d_q.p, d_home.p, nq, 1, d_bound2.p, d_neg2.p, d_out4.p, reinterpret_cast<uint64_t *>(d_scanned.p), nullptr) == RT_DP_OK,
              "knn angular tree: %s", rt_dp_last_error());
        CHECK(rt_dp_synchronize(ctx) == RT_DP_OK, "synchronize");
        {
            const auto e = d_neg.down(), t = d_neg2.down();
            CHECK(same_bits(e.data(), t.data(), e.size()) && d_out2.down() == d_out4.down(), "angular tree != angular exhaustive at n=%u", n);
            CHECK(d_scanned.down()[0] > 0, "no leaf scanned");
        }
        CHECK(rt_dp_knn_euclid16(ctx, d_pts.p + 1, n, d_q.p, nq, 0, 1, d_dist.p, d_out.p, nullptr) == RT_DP_EALIGN, "misaligned points accepted");
        CHECK(rt_dp_knn_euclid16(ctx, nullptr, n, d_q.p, nq, 0, 1, d_dist.p, d_out.p, nullptr) == RT_DP_ENULL, "NULL points accepted");
    }
}

static void device_trace(rt_dp_ctx *ctx) {
    Rng r(31);
    for (uint32_t n : {1u, 9u, 6000u}) {
        std::vector<float> tris((size_t)n * 9);
        for (uint32_t t = 0; t < n; ++t) {
            const float cx = r.f(-10.0f, 10.0f), cy = r.f(-10.0f, 10.0f), cz = r.f(5.0f, 30.0f);
            for (int v = 0; v < 3; ++v) { tris[(size_t)t * 9 + 3 * v] = cx + r.f(-1, 1); tris[(size_t)t * 9 + 3 * v + 1] = cy + r.f(-1, 1); tris[(size_t)t * 9 + 3 * v + 2] = cz + r.f(-1, 1); }
        }
        const uint32_t rows = rt_dp_bvh4_max_nodes(n), nr = 3000;
        std::vector<float> nodes((size_t)rows * 32), ot((size_t)n * 12);
        std::vector<uint32_t> order(n);
        uint32_t used = 0;
        CHECK(rt_dp_bvh4_build_host(tris.data(), n, 4, nodes.data(), &used, ot.data(), order.data()) == RT_DP_OK, "bvh build");
        std::vector<float> o[3], d[3];
        for (int k = 0; k < 3; ++k) { o[k].resize(nr); d[k].resize(nr); }
        for (uint32_t i = 0; i < nr; ++i) { o[0][i] = r.f(-8, 8); o[1][i] = r.f(-8, 8); o[2][i] = 0.0f; d[0][i] = r.f(-0.3f, 0.3f); d[1][i] = r.f(-0.3f, 0.3f); d[2][i] = 1.0f; }
        Dev<float> d_nodes(nodes), d_tris(ot), ox(o[0]), oy(o[1]), oz(o[2]), dx(d[0]), dy(d[1]), dz(d[2]), t0(nr), t1(nr);
        Dev<uint32_t> h0(nr), h1(nr);
        rt_dp_bvh4 bvh{sizeof(rt_dp_bvh4), used, n, 0, d_nodes.p, d_tris.p};
        const float *org[3] = {ox.p, oy.p, oz.p}, *dir[3] = {dx.p, dy.p, dz.p};
        uint64_t st[4];
        CHECK(rt_dp_trace_closest(ctx, &bvh, org, dir, nr, 4000.0f, 0, t0.p, h0.p, st) == RT_DP_OK, "depth-first walk n=%u: %s", n, rt_dp_last_error());
        CHECK(rt_dp_trace_closest(ctx, &bvh, org, dir, nr, 4000.0f, RT_DP_TRACE_WAVEFRONT, t1.p, h1.p, st) == RT_DP_OK, "wavefront walk n=%u: %s", n, rt_dp_last_error());
        const auto a = h0.down(), b = h1.down();
        uint32_t diff = 0, hits = 0;
        for (uint32_t i = 0; i < nr; ++i) { diff += a[i] != b[i]; hits += a[i] != 0xFFFFFFFFu; }
        CHECK(diff * 100 <= nr, "the two walks disagree on %u of %u rays", diff, nr);
        if (n >= 6000) CHECK(hits > 0, "no ray hit anything");
        // the hierarchy built on the device over the same triangles
        Dev<float> d_in(tris), b_nodes((size_t)rows * 32), b_tris((size_t)n * 12);
        Dev<uint32_t> b_order(n);
        uint32_t used2 = 0;
        CHECK(rt_dp_bvh4_build_device(ctx, d_in.p, n, 4, b_nodes.p, &used2, b_tris.p, b_order.p) == RT_DP_OK && used2 >= 1 && used2 <= rows, "bvh build on the device n=%u: %s", n, rt_dp_last_error());
        rt_dp_bvh4 bvh2{sizeof(rt_dp_bvh4), used2, n, 0, b_nodes.p, b_tris.p};
        CHECK(rt_dp_trace_closest(ctx, &bvh2, org, dir, nr, 4000.0f, 0, t1.p, h1.p, st) == RT_DP_OK, "walk over the device-built hierarchy: %s", rt_dp_last_error());
        const auto ta = t0.down(), tb = t1.down();
        uint32_t tdiff = 0;
        for (uint32_t i = 0; i < nr; ++i) tdiff += std::memcmp(&ta[i], &tb[i], 4) != 0;
        CHECK(tdiff * 100 <= nr, "closest t differs between the two hierarchies on %u of %u rays", tdiff, nr);
        // a corrupted hierarchy is refused on the device, not walked
        if (used > 1) {
            std::vector<float> broken = nodes;
            const uint32_t self = 0;  // child 0 of the root pointing at the root
            std::memcpy(&broken[24], &self, 4);
            Dev<float> d_broken(broken);
            rt_dp_bvh4 bad{sizeof(rt_dp_bvh4), used, n, 0, d_broken.p, d_tris.p};
            CHECK(rt_dp_trace_closest(ctx, &bad, org, dir, nr, 4000.0f, 0, t1.p, h1.p, st) == RT_DP_EINVAL, "a cyclic hierarchy was accepted");
        }
    }
}

static void device_streams_and_compaction(rt_dp_ctx *ctx) {
    const uint64_t n = 100000;
    const auto in = records(n, 4242);
    std::vector<rt_dp_output_bundle> host_out(n);
    CHECK(rt_dp_set_option(ctx, "stream_mode", 0) == RT_DP_OK, "stream_mode");
    CHECK(rt_dp_reset(ctx) == RT_DP_OK, "reset");
    CHECK(rt_dp_eval_bundles(ctx, in.data(), host_out.data(), n) == RT_DP_OK, "host stream: %s", rt_dp_last_error());
    Dev<rt_dp_input_bundle> d_in(in);
    Dev<rt_dp_output_bundle> d_out(n);
    for (int mode : {1, 2, 3}) {
        CHECK(rt_dp_set_option(ctx, "stream_mode", mode) == RT_DP_OK, "stream_mode");
        CHECK(rt_dp_reset(ctx) == RT_DP_OK, "reset");
        CHECK(rt_dp_eval_bundles_device(ctx, d_in.p, d_out.p, n) == RT_DP_OK, "device stream mode %d: %s", mode, rt_dp_last_error());
        const auto got = d_out.down();
        CHECK(std::memcmp(got.data(), host_out.data(), n * sizeof(rt_dp_output_bundle)) == 0, "device stream mode %d != host stream", mode);
    }
    CHECK(rt_dp_set_option(ctx, "stream_mode", 0) == RT_DP_OK, "stream_mode");

    // a Quadbox batch on the device, both layouts, and the compaction of its hit masks
    const uint64_t nb = 50001;
    const HostBatch h = make(nb, 0, 0, 0, 99);
    std::vector<Dev<float> *> ops;
    for (int f = 0; f < 34; ++f) ops.push_back(new Dev<float>(h.box[f]));
    Dev<float> t0(nb), t1(nb), t2(nb), t3(nb), ct(4 * nb), tiled_in(RT_DP_TILED_WORDS(nb, 34u));
    Dev<uint32_t> fl(nb), ci(4 * nb), cc(4 * nb), tiled_out(RT_DP_TILED_WORDS(nb, 5u));
    rt_dp_op_batch bin;
    rt_dp_result_batch bout;
    std::memset(&bin, 0, sizeof bin);
    std::memset(&bout, 0, sizeof bout);
    bin.struct_size = sizeof bin; bout.struct_size = sizeof bout;
    bin.flags = RT_DP_F_DEVICE_PTRS;
    bin.n[RT_DP_OP_QUADBOX] = nb;
    const float **bp = reinterpret_cast<const float **>(&bin.box);
    const void *arrays[34];
    for (int f = 0; f < 34; ++f) { bp[f] = ops[f]->p; arrays[f] = ops[f]->p; }
    bout.box.tmin_out[0] = t0.p; bout.box.tmin_out[1] = t1.p; bout.box.tmin_out[2] = t2.p; bout.box.tmin_out[3] = t3.p; bout.box.flags = fl.p;
    CHECK(rt_dp_eval_ctx(ctx, &bin, &bout) == RT_DP_OK, "device box batch: %s", rt_dp_last_error());
    uint64_t count = 0, count2 = 0;
    CHECK(rt_dp_box_compact(ctx, &bout.box, nb, nullptr, nullptr, ci.p, cc.p, ct.p, 4 * nb, &count) == RT_DP_OK, "box_compact: %s", rt_dp_last_error());
    uint64_t want = 0;
    for (uint32_t f : fl.down()) want += __builtin_popcount(f & 0xFu);
    CHECK(count == want, "compaction lists %llu entries, the hit masks hold %llu", (unsigned long long)count, (unsigned long long)want);
    CHECK(rt_dp_tile_pack(ctx, arrays, 34, nb, tiled_in.p, nullptr) == RT_DP_OK, "tile_pack: %s", rt_dp_last_error());
    bin.flags = RT_DP_F_DEVICE_PTRS | RT_DP_F_BOX_TILED;
    bin.box.org[0] = tiled_in.p;
    bout.box.tmin_out[0] = reinterpret_cast<float *>(tiled_out.p);
    CHECK(rt_dp_eval_ctx(ctx, &bin, &bout) == RT_DP_OK, "tiled device box batch: %s", rt_dp_last_error());
    Dev<uint32_t> ci2(4 * nb), cc2(4 * nb);
    Dev<float> ct2(4 * nb);
    CHECK(rt_dp_box_compact_tiled(ctx, tiled_out.p, nb, nullptr, nullptr, ci2.p, cc2.p, ct2.p, 4 * nb, &count2) == RT_DP_OK, "box_compact_tiled: %s", rt_dp_last_error());
    CHECK(count2 == count && ci.down() == ci2.down() && cc.down() == cc2.down(), "compaction of the tiled results differs");
    Dev<uint64_t> d_count(1);
    CHECK(rt_dp_box_compact_static(ctx, &bout.box, (uint64_t)RT_DP_BOX_COMPACT_STATIC_MAX_OPS + 1, nullptr, nullptr, ci.p, cc.p, ct.p, 4 * nb, d_count.p, nullptr) == RT_DP_EINVAL,
          "the allocation-free compaction accepted more ops than its array holds");
    for (auto *d : ops) delete d;
}

// The shard planners, several contexts driven from the plan, the kernel names, and the repair of a forced give-up.
static void sharding_and_giveups(rt_dp_ctx *ctx) {
    const HostBatch h = make(20000, 15000, 9000, 7000, 77);
    HostResults whole, parts;
    rt_dp_op_batch in; rt_dp_result_batch out;
    std::memset(&in, 0, sizeof in); std::memset(&out, 0, sizeof out);
    in.struct_size = sizeof in; out.struct_size = sizeof out;
    fill_plain(h, in);
    fill_results(h, whole, out);
    CHECK(rt_dp_reset(ctx) == RT_DP_OK && rt_dp_eval_ctx(ctx, &in, &out) == RT_DP_OK, "whole stream: %s", rt_dp_last_error());
    char name[128];
    for (int op = 0; op < 4; ++op) {
        CHECK(rt_dp_kernel_name(ctx, &in, &out, op, name, sizeof name) == RT_DP_OK && std::strncmp(name, "rtdp::k_", 8) == 0, "kernel name of opcode %d: '%s'", op, name);
    }
    CHECK(rt_dp_kernel_name(ctx, &in, &out, 7, name, sizeof name) == RT_DP_EOPCODE, "opcode 7 named");
    for (uint32_t world : {1u, 3u, 5u}) {
        std::vector<rt_dp_shard_range> plan(world);
        const uint64_t n[4] = {h.n_tri, h.n_box, h.n_euc, h.n_ang};
        CHECK(rt_dp_shard_plan(n, h.ectl.data(), h.actl.data(), world, 0, plan.data()) == RT_DP_OK, "shard plan: %s", rt_dp_last_error());
        rt_dp_result_batch pout;
        std::memset(&pout, 0, sizeof pout); pout.struct_size = sizeof pout;
        fill_results(h, parts, pout);
        for (uint32_t r = 0; r < world; ++r) {
            rt_dp_ctx *c = nullptr;
            CHECK(rt_dp_create(nullptr, &c) == RT_DP_OK, "create: %s", rt_dp_last_error());
            rt_dp_op_batch pin = in; rt_dp_result_batch po = pout;
            const uint64_t t0 = plan[r].start[0], b0 = plan[r].start[1], e0 = plan[r].start[2], a0 = plan[r].start[3];
            for (int op = 0; op < 4; ++op) pin.n[op] = plan[r].stop[op] - plan[r].start[op];
            for (int a = 0; a < 3; ++a) { pin.box.org[a] += b0; pin.box.dir[a] += b0; pin.box.inv[a] += b0; pin.tri.org[a] += t0; pin.tri.shear[a] += t0; }
            pin.box.extent += b0;
            for (int b = 0; b < 4; ++b) for (int f = 0; f < 6; ++f) pin.box.aabb[b][f] += b0;
            for (int v = 0; v < 3; ++v) for (int a = 0; a < 3; ++a) pin.tri.vtx[v][a] += t0;
            pin.tri.k += t0;
            pin.euc.a += e0 * 16; pin.euc.b += e0 * 16; pin.euc.ctl += e0; pin.ang.a += a0 * 8; pin.ang.b += a0 * 8; pin.ang.ctl += a0;
            for (int k = 0; k < 4; ++k) po.box.tmin_out[k] += b0;
            po.box.flags += b0; po.tri.t_num += t0; po.tri.t_denom += t0; po.tri.hit += t0;
            po.euc.accumulator += e0; po.ang.dot_product += a0; po.ang.norm += a0;
            CHECK(rt_dp_eval_ctx(c, &pin, &po) == RT_DP_OK, "part %u of %u: %s", r, world, rt_dp_last_error());
            rt_dp_destroy(c);
        }
        CHECK(same_bits(out.euc.accumulator, pout.euc.accumulator, h.n_euc) && same_bits(out.ang.dot_product, pout.ang.dot_product, h.n_ang) &&
              same_bits(out.ang.norm, pout.ang.norm, h.n_ang) && same_bits(out.box.tmin_out[2], pout.box.tmin_out[2], h.n_box) &&
              same_bits(out.tri.t_num, pout.tri.t_num, h.n_tri), "parts != whole stream at world %u", world);
    }
    CHECK(rt_dp_shard_plan(nullptr, nullptr, nullptr, 2, 0, nullptr) == RT_DP_ENULL, "NULL plan accepted");
    {
        const auto rec = records(5000, 91);
        std::vector<uint64_t> cuts(5);
        CHECK(rt_dp_shard_plan_records(rec.data(), rec.size(), 4, cuts.data()) == RT_DP_OK && cuts[0] == 0 && cuts[4] == rec.size(), "record plan: %s", rt_dp_last_error());
        CHECK(rt_dp_shard_plan_records(rec.data(), rec.size(), 0, cuts.data()) == RT_DP_EINVAL, "world 0 accepted");
    }
    // a forced give-up of the hand-over between workgroups: repaired by the synchronous entry point, same bits
    HostResults again;
    rt_dp_result_batch aout;
    std::memset(&aout, 0, sizeof aout); aout.struct_size = sizeof aout;
    fill_results(h, again, aout);
    CHECK(rt_dp_set_option(ctx, "debug_chain_skip", 1) == RT_DP_OK, "hook");
    CHECK(rt_dp_reset(ctx) == RT_DP_OK && rt_dp_eval_ctx(ctx, &in, &aout) == RT_DP_OK, "batch behind a give-up: %s", rt_dp_last_error());
    int fallbacks = 0;
    CHECK(rt_dp_get_option(ctx, "chain_fallbacks", &fallbacks) == RT_DP_OK && fallbacks >= 1, "no fallback counted (%d)", fallbacks);
    CHECK(same_bits(out.euc.accumulator, aout.euc.accumulator, h.n_euc) && same_bits(out.ang.dot_product, aout.ang.dot_product, h.n_ang), "repair changed bits");
    CHECK(rt_dp_set_option(ctx, "debug_chain_skip", -1) == RT_DP_OK, "hook off");
}

static void default_context_thread(int id) {
    const HostBatch h = make(1000 + id, 10, 100, 100, 40u + id);
    rt_dp_op_batch in;
    rt_dp_result_batch out;
    HostResults r;
    fill_plain(h, in);
    fill_results(h, r, out);
    for (int k = 0; k < 3; ++k)
        if (rt_dp_eval(&in, &out) != RT_DP_OK) { std::printf("FAIL thread %d: %s\n", id, rt_dp_last_error()); ++g_fail; }
}

int main() {
    std::setvbuf(stdout, nullptr, _IOLBF, 0);
    std::printf("%s, %d device(s)\n", rt_dp_version(), rt_dp_device_count());
    rt_dp_ctx *ctx = nullptr;
    if (rt_dp_create(nullptr, &ctx) != RT_DP_OK) { std::printf("rt_dp_create: %s\n", rt_dp_last_error()); return 2; }
    uint32_t seed = 1;
    for (uint64_t n : {0ull, 1ull, 2ull, 63ull, 64ull, 65ull, 255ull, 1000ull, 8191ull, 8192ull, 8193ull, 100003ull, 300000ull}) plain_and_tiled(ctx, n, seed++);
    std::printf("plain and tiled host batches done\n");
    streams(ctx);
    std::printf("record streams done\n");
    errors(ctx);
    std::printf("error paths done\n");
    builders();
    std::printf("host builders done\n");
    device_knn(ctx);
    std::printf("k-NN entry points done\n");
    device_trace(ctx);
    std::printf("traversal entry points done\n");
    device_streams_and_compaction(ctx);
    std::printf("device streams and compaction done\n");
    sharding_and_giveups(ctx);
    std::printf("shard plans, several contexts and give-up repair done\n");
    for (int k = 0; k < 20; ++k) {
        rt_dp_ctx *c = nullptr;
        CHECK(rt_dp_create(nullptr, &c) == RT_DP_OK, "create %d: %s", k, rt_dp_last_error());
        if (c && (k & 1)) plain_and_tiled(c, 500 + k, 900u + k);
        rt_dp_destroy(c);
    }
    std::printf("contexts done\n");
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t) th.emplace_back(default_context_thread, t);
        for (auto &t : th) t.join();
    }
    std::printf("threads done\n");
    rt_dp_destroy(ctx);
    std::printf(g_fail ? "host_san: %d FAILURES\n" : "host_san: ok\n", g_fail);
    std::fflush(stdout);
    // leave without the exit-time finalizers: with the HIP runtime linked into a sanitized executable, the sanitizer's
    // allocator asserts while the runtime unloads itself (its check, in its code, after everything of ours is done)
    std::_Exit(g_fail ? 1 : 0);
}
