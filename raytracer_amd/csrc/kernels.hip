// HIP kernels for the four UnifiedDatapath opcodes, written for gfx950 (CDNA4):
// 64-lane wavefronts, one lane owns whole ops (box, triangle) or a quarter /
// half of a beat (Euclidean / angular), 128-bit coalesced global accesses.
//
// Numerics contract (what makes the results bit-identical to the RTL's
// HardFloat units; reference external/hardfloat/src/main/scala/AddRecFN.scala,
// MulRecFN.scala, CompareRecFN.scala, RoundAnyRawFNToRecFN.scala):
//   * every add / sub / mul is a separate IEEE-754 binary32 RNE VALU op:
//     this file is built with -ffp-contract=off and carries the pragma below,
//     so no v_fma / v_mad / v_mac is ever formed;
//   * f32 subnormals are kept (the default FP mode of a HIP kernel on gfx9);
//   * min / max are NOT v_min / v_max: the RTL's compare-select keeps the first
//     operand when a >= b and yields the second one otherwise, also for NaN
//     (reference src/main/scala/raytracer_datapath/RecFNCompareSelect.scala:41-50);
//   * a NaN that reaches an output is rewritten to the RTL's canonical
//     0x7FC00000 (RoundAnyRawFNToRecFN.scala:249,277,280).
//
// File:line citations below are relative to the reference's
// src/main/scala/raytracer_datapath/UnifiedDatapath.scala unless stated.
#include "ops.hpp"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

#pragma clang fp contract(off)

namespace rtdp {


template <int V> struct Pack;
template <> struct Pack<1> { using F = float; using U = uint32_t; };
template <> struct Pack<2> { using F = f2v;   using U = u2v; };
template <> struct Pack<4> { using F = f4v;   using U = u4v; };

template <int V> __device__ __forceinline__ float lane_get(const typename Pack<V>::F &v, int i) { return v[i]; }
template <> __device__ __forceinline__ float lane_get<1>(const float &v, int) { return v; }
template <int V> __device__ __forceinline__ void lane_set(typename Pack<V>::F &v, int i, float x) { v[i] = x; }
template <> __device__ __forceinline__ void lane_set<1>(float &v, int, float x) { v = x; }
template <int V> __device__ __forceinline__ uint32_t lane_getu(const typename Pack<V>::U &v, int i) { return v[i]; }
template <> __device__ __forceinline__ uint32_t lane_getu<1>(const uint32_t &v, int) { return v; }
template <int V> __device__ __forceinline__ void lane_setu(typename Pack<V>::U &v, int i, uint32_t x) { v[i] = x; }
template <> __device__ __forceinline__ void lane_setu<1>(uint32_t &v, int, uint32_t x) { v = x; }

template <typename T, bool NT> __device__ __forceinline__ T ld(const void *p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const T *>(p));
    return *reinterpret_cast<const T *>(p);
}
template <typename T, bool NT> __device__ __forceinline__ void st(void *p, T v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<T *>(p));
    else *reinterpret_cast<T *>(p) = v;
}

template <int V, bool NT>
__device__ __forceinline__ void box_load(const BoxArgs &args, uint64_t i0, typename Pack<V>::F (&x)[kBoxFields]) {
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f) x[f] = ld<typename Pack<V>::F, NT>(args.in[f] + i0);
}

template <int V, bool NT>
__device__ __forceinline__ void box_eval_store(const BoxArgs &args, uint64_t i0, const typename Pack<V>::F (&x)[kBoxFields]) {
    using F = typename Pack<V>::F;
    using U = typename Pack<V>::U;
    F to[4];
    U fo;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        float r[kBoxFields], t[4];
        uint32_t fl;
#pragma unroll
        for (int f = 0; f < kBoxFields; ++f) r[f] = lane_get<V>(x[f], v);
        box_one(r, t, fl);
#pragma unroll
        for (int k = 0; k < 4; ++k) lane_set<V>(to[k], v, t[k]);
        lane_setu<V>(fo, v, fl);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) st<F, NT>(args.tmin[k] + i0, to[k]);
    st<U, NT>(args.flags + i0, fo);
}

// One-shot shape: a lane owns V consecutive ops, issues all of its loads, then computes and stores.
template <int V, bool NT>
__global__ __launch_bounds__(256) void k_box(BoxArgs args, uint64_t first, uint64_t groups) {
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= groups) return;
    const uint64_t i0 = first + g * V;
    typename Pack<V>::F x[kBoxFields];
    box_load<V, NT>(args, i0, x);
    box_eval_store<V, NT>(args, i0, x);
}

// The same with the ray's reciprocals derived in the kernel: inv = 1.0f / dir, correctly rounded, as the
// reference's software model fills the field (src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:56-61).
// Three operand arrays fewer: 144 B per op instead of 156.
template <int V, bool NT>
__global__ __launch_bounds__(256) void k_box_derive(BoxArgs args, uint64_t first, uint64_t groups) {
    using F = typename Pack<V>::F;
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= groups) return;
    const uint64_t i0 = first + g * V;
    F x[kBoxFields];
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f)
        if (f < 6 || f > 8) x[f] = ld<F, NT>(args.in[f] + i0);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) lane_set<V>(x[6 + c], v, canon_nan(1.0f / lane_get<V>(x[3 + c], v)));
    box_eval_store<V, NT>(args, i0, x);
}

// Tiled operand layout: [tile][field][kTileOps] with whole tiles, i.e. every field of kTileOps consecutive ops is one
// contiguous 32 KiB run and a tile's runs follow each other.  The plain layout streams 39 arrays in lockstep, whose
// mutual spacing (a caller's choice) decides how often their cache lines contend for one channel; here the spacing
// is fixed by the layout.  Measured on MI355X against tile size: 64 .. 256 ops per tile are slower than the plain
// layout (too short a run per DRAM page visit), 2048 .. 16384 are faster at every batch size, 8192 is the best.
template <bool NT, bool DERIVE>
__global__ __launch_bounds__(256) void k_box_tiled(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t tile = i / kTileOps;
    const uint32_t w = (uint32_t)(i % kTileOps);
    const float *p = in + tile * (uint64_t)(kBoxFields * kTileOps) + w;
    float x[kBoxFields];
    // How a wave hands its 34 loads to the memory pipeline matters (measured, 2^22 .. 2^26 ops, 3 workgroups per CU,
    // fractions of the HBM peak): all loads back to back from a scalar tile base with one shared lane offset 0.82;
    // per-lane 64-bit addresses (two vector adds between loads) 0.845; a few idle cycles after each load 0.855 .. 0.86,
    // flat from 2 to 10 cycles, falling off from 16 (0.83, 0.75 at 32).  Waiting for the first one or two loads
    // before the rest go out gives the same 0.855 .. 0.86: what helps is that the waves of a CU do not all push their
    // bursts at once.  (With the idle cycles in place the compiler also tests the ray direction as soon as it has
    // arrived, i.e. this kernel does both.)  The triangle kernel's 16 loads gain nothing from either.
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f) {
        if (DERIVE && f >= 6 && f <= 8) continue;
        x[f] = ld<float, NT>(p + (size_t)f * kTileOps);
        asm volatile("s_nop 5");
    }
    if (DERIVE) {
#pragma unroll
        for (int c = 0; c < 3; ++c) x[6 + c] = canon_nan(1.0f / x[3 + c]);
    }
    float t[4];
    uint32_t fl;
    box_one(x, t, fl);
    uint32_t *q = out + tile * (uint64_t)(kBoxOutWords * kTileOps) + w;
#pragma unroll
    for (int k = 0; k < 4; ++k) st<uint32_t, NT>(q + (size_t)k * kTileOps, __float_as_uint(t[k]));
    st<uint32_t, NT>(q + (size_t)4 * kTileOps, fl);
}

// Diagnostic shape for bandwidth accounting (variant 10, never selected by default): the
// box kernel's loads and stores with a trivial reduction in place of the datapath
// arithmetic, i.e. what the memory pattern alone costs.  Its outputs are meaningless.
template <int V>
__global__ __launch_bounds__(256) void k_box_pattern(BoxArgs args, uint64_t groups) {
    using F = typename Pack<V>::F;
    using U = typename Pack<V>::U;
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= groups) return;
    const uint64_t i0 = g * V;
    F x[kBoxFields];
    box_load<V, true>(args, i0, x);
    F acc = x[0];
#pragma unroll
    for (int f = 1; f < kBoxFields; ++f) acc += x[f];
    U fo;
#pragma unroll
    for (int v = 0; v < V; ++v) lane_setu<V>(fo, v, __float_as_uint(lane_get<V>(acc, v)));
#pragma unroll
    for (int k = 0; k < 4; ++k) st<F, true>(args.tmin[k] + i0, acc);
    st<U, true>(args.flags + i0, fo);
}

template <int V, bool NT>
__device__ __forceinline__ void tri_load(const TriArgs &args, uint64_t i0, typename Pack<V>::F (&x)[kTriFields], typename Pack<V>::U &kk) {
#pragma unroll
    for (int f = 0; f < kTriFields; ++f) x[f] = ld<typename Pack<V>::F, NT>(args.in[f] + i0);
    kk = ld<typename Pack<V>::U, NT>(args.k + i0);
}

template <int V, bool NT>
__device__ __forceinline__ void tri_eval_store(const TriArgs &args, uint64_t i0, const typename Pack<V>::F (&x)[kTriFields], const typename Pack<V>::U &kk) {
    using F = typename Pack<V>::F;
    using U = typename Pack<V>::U;
    F on, od;
    U oh;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        float r[kTriFields], n, d;
        uint32_t h;
#pragma unroll
        for (int f = 0; f < kTriFields; ++f) r[f] = lane_get<V>(x[f], v);
        tri_one(r, lane_getu<V>(kk, v), n, d, h);
        lane_set<V>(on, v, n);
        lane_set<V>(od, v, d);
        lane_setu<V>(oh, v, h);
    }
    st<F, NT>(args.t_num + i0, on);
    st<F, NT>(args.t_denom + i0, od);
    st<U, NT>(args.hit + i0, oh);
}

template <int V, bool NT>
__global__ __launch_bounds__(256) void k_tri(TriArgs args, uint64_t first, uint64_t groups) {
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= groups) return;
    const uint64_t i0 = first + g * V;
    typename Pack<V>::F x[kTriFields];
    typename Pack<V>::U kk;
    tri_load<V, NT>(args, i0, x, kk);
    tri_eval_store<V, NT>(args, i0, x, kk);
}

// The triangle op over the tiled layout: 16 words in per op (15 operands and the packed selectors), 3 words out.
template <bool NT, bool DERIVE>
__global__ __launch_bounds__(256) void k_tri_tiled(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t tile = i / kTileOps;
    const uint32_t w = (uint32_t)(i % kTileOps);
    const float *p = in + tile * (uint64_t)(kTriInWords * kTileOps) + w;
    float x[kTriFields];
#pragma unroll
    for (int f = 0; f < kTriFields; ++f) x[f] = ld<float, NT>(p + (size_t)f * kTileOps);
    uint32_t kk;
    if (DERIVE) {
        const float d[3] = {x[3], x[4], x[5]};
        float inv[3], sh[3];
        ray_derive(d, inv, kk, sh);
#pragma unroll
        for (int c = 0; c < 3; ++c) x[3 + c] = sh[c];
    } else {
        kk = ld<uint32_t, NT>(p + (size_t)kTriFields * kTileOps);
    }
    float tn, td;
    uint32_t h;
    tri_one(x, kk, tn, td, h);
    uint32_t *q = out + tile * (uint64_t)(kTriOutWords * kTileOps) + w;
    st<uint32_t, NT>(q, __float_as_uint(tn));
    st<uint32_t, NT>(q + (size_t)kTileOps, __float_as_uint(td));
    st<uint32_t, NT>(q + (size_t)2 * kTileOps, h);
}

// Plain arrays <-> tiled buffer, any number of 32-bit arrays up to kTileMaxArrays.
//
// A tile's run of one word is kTileOps contiguous words in the tiled buffer AND in the plain array, so the conversion is
// a set of contiguous 32 KiB copies with 16-byte accesses.  A workgroup moves a quarter of a run (8 KiB: a lane's two
// loads are issued before its first store); the (tile, word, quarter) index runs quarter-fastest, then word, so the
// workgroups in flight write (pack) or read (unpack) the tiled buffer as one sequential stream.  Each byte is touched
// once: nt hints on both sides.  Measured on MI355X, 34 arrays, fractions of the HBM peak over bytes read + written
// (2^22 / 2^24 / 2^26 ops): whole runs per workgroup 0.72 at 2^24 whatever the residency; halves 0.775 (4 workgroups per
// CU); quarters with 3 workgroups per CU 0.78 / 0.78 / 0.76; eighths 0.77 / 0.80 / 0.69 uncapped; tile-fastest order
// 0.66 - 0.74.  (A first version guarded every load by the tile's live count: the compiler then waited for each load
// before it issued the next, 0.53.)
constexpr int kCopySplit = 4, kCopyWgsPerCu = 3;

template <bool PACK>
__global__ __launch_bounds__(256) void k_tile_copy16(TileList l, uint32_t count, uint64_t n, uint32_t *__restrict__ tiled) {
    constexpr int kChunks = kTileOps / 4 / 256 / kCopySplit;  // 16-byte chunks per lane: 2
    const uint32_t part = blockIdx.x % kCopySplit, b = blockIdx.x / kCopySplit;
    const uint64_t tile = b / count;
    const uint32_t w = b % count;
    uint32_t *__restrict__ plain = l.a[w] + tile * kTileOps;
    uint32_t *__restrict__ run = tiled + (tile * count + w) * (uint64_t)kTileOps;
    const uint64_t left = n - tile * kTileOps;
    const uint32_t *src = PACK ? plain : run;
    uint32_t *dst = PACK ? run : plain;
    if (left >= kTileOps) {  // a whole tile (uniform branch): unconditional loads, all in flight before the first store
        u4v v[kChunks];
#pragma unroll
        for (int k = 0; k < kChunks; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(src) + (part * kChunks + k) * 256 + threadIdx.x);
#pragma unroll
        for (int k = 0; k < kChunks; ++k) __builtin_nontemporal_store(v[k], reinterpret_cast<u4v *>(dst) + (part * kChunks + k) * 256 + threadIdx.x);
        return;
    }
    // the last, partial tile: what lies behind its ops is neither read nor written
    if (part == 0)
        for (uint32_t i = threadIdx.x; i < (uint32_t)left; i += 256u) dst[i] = src[i];
}

// The same for arrays that are not 16-byte aligned: one lane per op, word by word.
__global__ __launch_bounds__(256) void k_tile_pack(TileList l, int count, uint64_t n, uint32_t *__restrict__ tiled) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t *q = tiled + (i / kTileOps) * (uint64_t)count * kTileOps + (i % kTileOps);
    for (int f = 0; f < count; ++f) q[(size_t)f * kTileOps] = l.a[f][i];
}

__global__ __launch_bounds__(256) void k_tile_unpack(const uint32_t *__restrict__ tiled, int count, uint64_t n, TileList l) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t *q = tiled + (i / kTileOps) * (uint64_t)count * kTileOps + (i % kTileOps);
    for (int f = 0; f < count; ++f) l.a[f][i] = q[(size_t)f * kTileOps];
}

// The same with shear and the axis selectors derived in the kernel from the ray direction, which then
// takes the place of shear in fields 3..5 (ray_derive): no selector array, 72 B per op instead of 76.
template <int V, bool NT>
__global__ __launch_bounds__(256) void k_tri_derive(TriArgs args, uint64_t first, uint64_t groups) {
    using F = typename Pack<V>::F;
    using U = typename Pack<V>::U;
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= groups) return;
    const uint64_t i0 = first + g * V;
    F x[kTriFields];
    U kk;
#pragma unroll
    for (int f = 0; f < kTriFields; ++f) x[f] = ld<F, NT>(args.in[f] + i0);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const float d[3] = {lane_get<V>(x[3], v), lane_get<V>(x[4], v), lane_get<V>(x[5], v)};
        float inv[3], sh[3];
        uint32_t k;
        ray_derive(d, inv, k, sh);
#pragma unroll
        for (int c = 0; c < 3; ++c) lane_set<V>(x[3 + c], v, sh[c]);
        lane_setu<V>(kk, v, k);
    }
    tri_eval_store<V, NT>(args, i0, x, kk);
}

// ------------------------------------------------- Euclidean / angular beats
// A beat's 16 (8) element pairs are spread over 4 (2) adjacent lanes, four
// elements per lane, so one wave-instruction loads 1 KiB of contiguous vector
// data.  The RTL's adder tree pairs element i with i+8, i+4, i+2, i+1
// (:632-649, :783-803, :915-935, :1007-1027; angular: i+4, i+2, i+1, :650-676,
// :804-832, :936-964): the first levels are lane-xor exchanges inside the quad
// (DPP quad_perm, no LDS), the last two are in-lane adds.  IEEE addition is
// commutative, so both partners of an exchange hold the same bits.
//
// The accumulator (:1133-1154, angular :1028-1064) is a strictly sequential
// float sum over the beats of one vector pair, so it cannot be re-associated.
// Two kernel shapes respect that, with identical results:
//   k_vec ("owning")        a workgroup owns whole vector pairs: its range starts at the first pair boundary at
//                           or after blockIdx * tile and ends at the first pair boundary at or after
//                           (blockIdx + 1) * tile; a pair longer than a tile is handled entirely by the
//                           workgroup in whose nominal range it starts;
//   k_vec_chain ("chained") fixed tiles; the pair that enters a tile from the previous one is continued from
//                           the running sums the previous workgroup publishes (further down).
// In both, inside a sub-tile (PASSES x 64 or x 128 beats) the per-beat tree sums go to LDS and one lane per
// vector pair walks its beats in order.

constexpr int kThreads = 256;  // threads per workgroup of the beat kernels

__device__ __forceinline__ float quad_xor1(float v) {  // quad_perm [1,0,3,2]
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {  // quad_perm [2,3,0,1]
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}

// First pair boundary (a beat index whose predecessor carries the reset bit; 0 and n count as boundaries) at or
// after `nominal`, looked for in windows of 64 beats that start below `limit`; a value >= limit means that no
// boundary lies in [nominal, limit).  Executed by one full wave; at most (limit - nominal) / 64 + 1 loads per lane.
__device__ __forceinline__ uint64_t pair_boundary(const uint32_t *ctl, uint64_t nominal, uint64_t limit, uint64_t n, int lane, uint32_t probe) {
    if (nominal == 0) return 0;
    if (nominal >= n) return n;
    {   // first window: the control words were fetched earlier (pair_probe), so that other loads could be issued meanwhile
        const uint64_t i = nominal + lane;
        const bool p = (i < n) ? (((probe >> 16) & 1u) != 0u) : true;
        const unsigned long long m = __ballot(p);
        if (m) {
            const uint64_t r = nominal + (uint64_t)(__ffsll((long long)m) - 1);
            return r < n ? r : n;
        }
    }
    for (uint64_t i0 = nominal + 64; i0 < limit; i0 += 64) {
        const uint64_t i = i0 + lane;
        const bool p = (i < n) ? (((ctl[i - 1] >> 16) & 1u) != 0u) : true;  // indices >= n act as boundaries
        const unsigned long long m = __ballot(p);
        if (m) {
            const uint64_t r = i0 + (uint64_t)(__ffsll((long long)m) - 1);
            return r < n ? r : n;
        }
    }
    return limit < n ? limit : n;  // not a boundary: only "none below limit"
}

// The control word pair_boundary's first window needs from this lane (issued early, consumed later).
__device__ __forceinline__ uint32_t pair_probe(const uint32_t *ctl, uint64_t nominal, uint64_t n, int lane) {
    const uint64_t i = nominal + lane;
    return (nominal != 0 && i < n) ? ctl[i - 1] : 0u;
}

// The same without a limit, for the end of a vector pair that runs on beyond the tile after the one it started
// in: 512 beats per round with eight loads in flight per lane, so the search costs a small fraction of the serial
// accumulation the same workgroup then performs over those beats.  Terminates at n.
__device__ __forceinline__ uint64_t pair_boundary_far(const uint32_t *ctl, uint64_t start, uint64_t n, int lane) {
    if (start >= n) return n;
    for (uint64_t i0 = start;; i0 += 512) {
        bool p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint64_t i = i0 + (uint64_t)k * 64 + lane;
            p[k] = (i < n) ? (((ctl[i - 1] >> 16) & 1u) != 0u) : true;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long m = __ballot(p[k]);
            if (m) {
                const uint64_t r = i0 + (uint64_t)k * 64 + (uint64_t)(__ffsll((long long)m) - 1);
                return r < n ? r : n;
            }
        }
    }
}

// A value every lane of the wave holds alike (read from LDS, hence in vector registers) moved to scalar registers:
// the tile bounds and loop counters derived from it then cost no vector registers.
__device__ __forceinline__ uint64_t wave_uniform(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Register budget: the angular shape with four load pairs per lane must stay within 72 VGPRs (7 waves per SIMD, 7
// resident workgroups per CU); one register more drops it to 6 and costs about 5 % (measured), and squeezing it into
// 64 for 8 waves makes the allocator spill inside the beat loop (0.56 of the HBM roof instead of 0.75).
template <bool EUC, bool NT, int PASSES>
__global__ __launch_bounds__(256, (PASSES <= 4) ? 7 : 4) void k_vec(VecArgs args, uint64_t n, uint32_t tile, int walk_mode) {
    constexpr int kLook = 4;                 // beat sums fetched per LDS round trip by the long-pair walk
    constexpr int LPB = EUC ? 4 : 2;         // lanes per beat
    constexpr int BPP = kThreads / LPB;      // beats per pass
    constexpr int SUB = PASSES * BPP;        // beats per sub-tile
    constexpr int ROW = EUC ? 16 : 8;        // floats per beat and vector

    __shared__ float s_sum0[SUB + kLook];
    __shared__ float s_sum1[EUC ? 1 : SUB + kLook];
    __shared__ uint32_t s_ctl[SUB + kLook];
    __shared__ float s_out0[SUB + kLook];
    __shared__ float s_out1[EUC ? 1 : SUB + kLook];
    __shared__ uint64_t s_range[2];
    __shared__ float s_carry[2][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // This workgroup owns the vector pairs that START in its nominal range [first, next): its beats run from lo, the
    // first pair boundary at or after first, to hi, the first pair boundary at or after next.  The operand loads of
    // the first sub-tile are issued from `first` before lo and hi are known, so that the boundary searches (loads of
    // control words, one memory round trip) overlap with them instead of preceding them; the few beats in front of
    // lo belong to the previous workgroup's last pair and are dropped.  Both searches are limited to one tile: a
    // workgroup whose range holds no pair start leaves after tile / 64 loads per lane however long the pair that
    // covers it is, and only a workgroup that owns a long pair looks further for its end -- over beats it then has
    // to accumulate one by one anyway.
    //
    // The overlap needs registers for the searches while the operand loads are in flight; the shapes with eight load
    // pairs per lane run at four waves per SIMD and have them, the shapes that must stay within 72 registers would
    // spill (measured on the angular default: 0.65 of the HBM roof instead of 0.75) and search first, then load from lo.
    constexpr bool kOverlap = (PASSES >= 8);
    const uint64_t first = (uint64_t)blockIdx.x * tile, next = first + tile;
    const int part = tid % LPB;  // which four elements of the beat
    const uint32_t probe = (wave < 2) ? pair_probe(args.ctl, wave ? next : first, n, lane) : 0u;
    uint64_t lo = first, hi = n;
    bool long_pair = false;

    // lo, hi and the kind of walk from the two searches; false: no pair starts in this workgroup's range
    auto search = [&]() {
        if (wave < 2) {
            const uint64_t from = wave ? next : first;
            const uint64_t r = pair_boundary(args.ctl, from, from + tile, n, lane, probe);
            if (lane == 0) s_range[wave] = r;
        }
        if (tid == 0) {
            // only the workgroup that starts at beat 0 can inherit a running sum
            s_carry[0][0] = (blockIdx.x == 0) ? args.state_in[0] : 0.0f;
            s_carry[0][1] = (!EUC && blockIdx.x == 0) ? args.state_in[1] : 0.0f;
        }
    };
    auto resolve = [&]() -> bool {   // behind a barrier that follows search()
        lo = wave_uniform(s_range[0]);
        if (lo >= next || lo >= n) return false;
        hi = wave_uniform(s_range[1]);
        if (hi < n && hi >= next + tile) {  // the last pair that starts here runs on beyond the next tile (uniform branch)
            __syncthreads();
            if (wave == 0) {
                const uint64_t r = pair_boundary_far(args.ctl, hi, n, lane);
                if (lane == 0) s_range[1] = r;
            }
            __syncthreads();
            hi = wave_uniform(s_range[1]);
        }
        // a vector pair of more than four tiles is in this range (uniform); variant 21 / 22 force one walk or the other (A/B)
        long_pair = (walk_mode == 0) ? (hi - lo > 4ull * tile + (uint64_t)tile) : (walk_mode == 2);
        return true;
    };
    if (!kOverlap) {
        search();
        __syncthreads();
        if (!resolve()) return;
    }
    const uint64_t start = kOverlap ? first : lo;
    int par = 0;
    for (uint64_t base = start;; base += SUB, par ^= 1) {
        const bool head = (base == start);            // first sub-tile
        const bool pending = kOverlap && head;        // ... whose lo and hi are still being looked for
        const uint64_t left = (pending ? n : hi) - base;
        const int loaded = (left < (uint64_t)SUB) ? (int)left : SUB;

        // phase 1: tree sums of this sub-tile's beats
        f4v va[PASSES], vb[PASSES];
        uint32_t vc[PASSES];
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            const int bl = u * BPP + tid / LPB;
            if (bl < loaded) {
                const uint64_t e = (base + bl) * ROW + part * 4;
                va[u] = ld<f4v, NT>(args.a + e);
                vb[u] = ld<f4v, NT>(args.b + e);
                vc[u] = args.ctl[base + bl];
            } else {
                va[u] = (f4v)(0.0f); vb[u] = (f4v)(0.0f); vc[u] = 0u;
            }
        }
        if (pending) search();
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            const int bl = u * BPP + tid / LPB;
            const uint32_t mbits = vc[u] >> (part * 4);
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = (mbits >> j) & 1u;
                if (EUC) {
                    const float d = on ? (va[u][j] - vb[u][j]) : 0.0f;  // :221-245
                    s0[j] = d * d;                                      // :399-419
                } else {
                    const float q = on ? va[u][j] : 0.0f;               // :246-268
                    const float c = on ? vb[u][j] : 0.0f;
                    s0[j] = q * c;                                      // :420-453
                    s1[j] = c * c;
                }
            }
            float r0, r1 = 0.0f;
            if (EUC) {
#pragma unroll
                for (int j = 0; j < 4; ++j) s0[j] = s0[j] + quad_xor2(s0[j]);  // i, i+8
#pragma unroll
                for (int j = 0; j < 4; ++j) s0[j] = s0[j] + quad_xor1(s0[j]);  // i, i+4
                r0 = (s0[0] + s0[2]) + (s0[1] + s0[3]);                        // i, i+2 then 0, 1
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { s0[j] = s0[j] + quad_xor1(s0[j]); s1[j] = s1[j] + quad_xor1(s1[j]); }  // i, i+4
                r0 = (s0[0] + s0[2]) + (s0[1] + s0[3]);
                r1 = (s1[0] + s1[2]) + (s1[1] + s1[3]);
            }
            if (part == 0 && bl < loaded) {
                s_sum0[bl] = r0;
                if (!EUC) s_sum1[bl] = r1;
                s_ctl[bl] = vc[u];
            }
        }
        __syncthreads();
        if (pending && !resolve()) return;  // no pair starts here
        // beats of this sub-tile that are ours: [own, cnt); own > 0 only in front of lo
        const int cnt = (hi - base < (uint64_t)loaded) ? (int)(hi - base) : loaded;
        const int own = (lo > base) ? ((lo - base < (uint64_t)cnt) ? (int)(lo - base) : cnt) : 0;
        // the register entering beat 0 of the sub-tile is the carry of the previous one when a pair runs on across
        // the border, and what the stream handed over (s_carry[0], set above) when the first sub-tile starts a pair
        const bool carry_in = (base > lo) || head;

        // phase 2: one lane per vector pair accumulates its beats in order.  Ordinary streams (pairs of a few beats)
        // take the plain loop: one beat per LDS round trip, least instructions, and the round trips hide behind the
        // other workgroups of the CU.  A workgroup that owns a long pair is alone with a serial chain; there the
        // look-ahead form fetches four beat sums per round trip (the arrays are padded, so the look-ahead past a
        // pair's end or past cnt stays in bounds and is simply not used).  The choice is uniform over the workgroup.
        if (!long_pair) {
            for (int t0 = tid; t0 < cnt; t0 += kThreads) {
                const bool starts = (t0 >= own) && ((t0 == 0) || (((s_ctl[t0 - 1] >> 16) & 1u) != 0u));
                if (starts) {
                    float acc0 = (t0 == 0 && carry_in) ? s_carry[par][0] : 0.0f;
                    float acc1 = (t0 == 0 && carry_in) ? s_carry[par][1] : 0.0f;
                    int i = t0;
                    bool rs;
                    // the loop body is the whole cost of this phase (few lanes are active, every wave pays for the longest
                    // pair it holds), so it carries nothing that can wait: NaN results are made canonical when the sub-tile
                    // is stored, and the reset only matters to what is handed on
                    do {
                        acc0 = acc0 + s_sum0[i];
                        s_out0[i] = acc0;
                        rs = ((s_ctl[i] >> 16) & 1u) != 0u;
                        if (!EUC) {
                            acc1 = acc1 + s_sum1[i];
                            s_out1[i] = acc1;
                        }
                        ++i;
                    } while (i < cnt && !rs);
                    if (i == cnt) {  // exactly one lane reaches the end of the sub-tile
                        s_carry[par ^ 1][0] = rs ? 0.0f : acc0;
                        s_carry[par ^ 1][1] = rs ? 0.0f : acc1;
                    }
                }
            }
        } else {
            for (int t0 = tid; t0 < cnt; t0 += kThreads) {
                const bool starts = (t0 >= own) && ((t0 == 0) || (((s_ctl[t0 - 1] >> 16) & 1u) != 0u));
                if (starts) {
                    float acc0 = (t0 == 0 && carry_in) ? s_carry[par][0] : 0.0f;
                    float acc1 = (t0 == 0 && carry_in) ? s_carry[par][1] : 0.0f;
                    int i = t0;
                    bool more = true;
                    while (more) {
                        float v0[kLook], v1[kLook];
                        uint32_t c[kLook];
#pragma unroll
                        for (int k = 0; k < kLook; ++k) {
                            v0[k] = s_sum0[i + k];
                            if (!EUC) v1[k] = s_sum1[i + k];
                            c[k] = s_ctl[i + k];
                        }
#pragma unroll
                        for (int k = 0; k < kLook; ++k) {
                            if (more) {
                                const bool rs = ((c[k] >> 16) & 1u) != 0u;
                                const float o0 = acc0 + v0[k];
                                s_out0[i + k] = o0;
                                acc0 = rs ? 0.0f : o0;
                                if (!EUC) {
                                    const float o1 = acc1 + v1[k];
                                    s_out1[i + k] = o1;
                                    acc1 = rs ? 0.0f : o1;
                                }
                                const bool last = (i + k + 1 == cnt);
                                if (last) {
                                    s_carry[par ^ 1][0] = acc0;
                                    s_carry[par ^ 1][1] = acc1;
                                }
                                more = !rs && !last;
                            }
                        }
                        i += kLook;
                    }
                }
            }
        }
        __syncthreads();

        // phase 3: coalesced result store
        for (int t0 = tid; t0 < cnt; t0 += kThreads) {
            if (t0 >= own) {
                st<float, NT>(args.out0 + base + t0, canon_nan(s_out0[t0]));
                if (!EUC) st<float, NT>(args.out1 + base + t0, canon_nan(s_out1[t0]));
            }
        }
        if (base + SUB >= hi) break;
    }
    if (hi == n && tid == 0) {  // the workgroup that owns the last beat hands the registers on (written by its last walk)
        args.state_out[0] = canon_nan(s_carry[par ^ 1][0]);
        if (!EUC) args.state_out[1] = canon_nan(s_carry[par ^ 1][1]);
    }
}

// ---- the chained shape -----------------------------------------------------------------------------------------
// Fixed tiles: workgroup b evaluates beats [b * SUB, (b + 1) * SUB) and nothing else -- no search for pair
// boundaries, no second sub-tile for the pair that hangs over the end, no operand loaded twice.  The one pair that
// enters a tile from the previous one is continued from the running sums the previous workgroup publishes when its
// walk reaches the end of its tile: one 64-bit word per sum, {launch epoch, float bits}, written and polled with
// device-scope atomics, so a word is valid exactly when it carries this launch's epoch and nothing has to be cleared
// between launches.  Only the lane that continues that pair waits; every other pair of the tile starts from +0 and is
// walked meanwhile.  The arithmetic is the same strictly sequential sum in the same order (the continuing lane adds
// its beats one by one onto the value it received), so the results do not depend on the shape.
//
// Progress: a workgroup waits only for the workgroup with the next lower index, which was dispatched before it and
// waits, if at all, for a still lower one; workgroup 0 never waits.  A pair spanning many tiles is a serial chain by
// definition of the arithmetic and costs one walk per tile, one after the other.  The wait is bounded all the same
// (a lost workgroup must not hang the device, and dispatch in index order is what the hardware does, not something
// the programming model promises): after kChainSpins polls the lane counts a give-up in the header word and goes on
// with +0.  The host reads that word behind every launch it waits for and re-evaluates the segment in the owning
// shape, which needs no hand-over, when it moved (api.cpp), so a give-up costs time, never a wrong bit.
constexpr int kChainSpins = 1 << 20;  // about a second; the longest legitimate wait is tens of milliseconds
constexpr int kChainSpinsDebug = 1 << 12;

__device__ __forceinline__ unsigned long long chain_word(uint32_t epoch, float v) {
    return ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v);
}

__device__ __forceinline__ float chain_wait(unsigned long long *slot, uint32_t epoch, unsigned long long *header, int spins) {
    for (int spin = 0; spin < spins; ++spin) {
        const unsigned long long w = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(w >> 32) == epoch) return __uint_as_float((uint32_t)w);
        __builtin_amdgcn_s_sleep(4);
    }
    atomicAdd(header, 1ull);
    return 0.0f;
}

template <int EPL> struct VecLoad;
template <> struct VecLoad<4> { using T = f4v; };
template <> struct VecLoad<2> { using T = f2v; };

// EPL: elements of a beat per lane = width of the operand loads (4: 16-byte loads; 2: 8-byte loads, with which the
// angular beat spreads over four lanes like the Euclidean one and its adder tree gains the same lane-xor-2 level in
// front -- measured no faster, 0.746 against 0.743 of the HBM roof, and not instantiated).
template <bool EUC, bool NT, int PASSES, int EPL>
__global__ __launch_bounds__(256, (PASSES * EPL <= 16) ? 7 : 4) void k_vec_chain(VecArgs args, uint64_t n) {
    static_assert(EPL == 4 || (EPL == 2 && !EUC), "the Euclidean beat needs all four lanes of a quad at EPL = 4");
    using VT = typename VecLoad<EPL>::T;
    constexpr int kLook = 4;                 // beat sums fetched per LDS round trip by the sparse-reset walk
    constexpr int ROW = EUC ? 16 : 8;        // floats per beat and vector
    constexpr int LPB = ROW / EPL;           // lanes per beat
    constexpr int BPP = kThreads / LPB;      // beats per pass
    constexpr int SUB = PASSES * BPP;        // beats per tile

    __shared__ float s_sum0[SUB + kLook];
    __shared__ float s_sum1[EUC ? 1 : SUB + kLook];
    __shared__ uint32_t s_ctl[SUB + kLook];
    __shared__ float s_out0[SUB + kLook];
    __shared__ float s_out1[EUC ? 1 : SUB + kLook];

    const int tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * SUB;
    const uint64_t left = n - base;
    const int cnt = (left < (uint64_t)SUB) ? (int)left : SUB;
    const bool last_tile = (left <= (uint64_t)SUB);
    const int part = tid % LPB;  // which EPL elements of the beat
    unsigned long long *const header = args.chain_header;
    unsigned long long *const slots = args.chain;  // two words per tile
    const int spins = args.debug_skip ? kChainSpinsDebug : kChainSpins;

    // phase 1: tree sums of the tile's beats
    VT va[PASSES], vb[PASSES];
    uint32_t vc[PASSES];
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
        const int bl = u * BPP + tid / LPB;
        if (bl < cnt) {
            const uint64_t e = (base + bl) * ROW + part * EPL;
            va[u] = ld<VT, NT>(args.a + e);
            vb[u] = ld<VT, NT>(args.b + e);
            vc[u] = args.ctl[base + bl];
        } else {
            va[u] = (VT)(0.0f); vb[u] = (VT)(0.0f); vc[u] = 0u;
        }
    }
    // does a pair enter this tile from the previous one?  (thread 0 only; the word arrives with the operands)
    const uint32_t before = (tid == 0 && base != 0) ? args.ctl[base - 1] : 0x10000u;
    bool any_reset = false;
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
        const int bl = u * BPP + tid / LPB;
        const uint32_t mbits = vc[u] >> (part * EPL);
        float s0[EPL], s1[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const bool on = (mbits >> j) & 1u;
            if (EUC) {
                const float d = on ? (va[u][j] - vb[u][j]) : 0.0f;  // :221-245
                s0[j] = d * d;                                      // :399-419
            } else {
                const float q = on ? va[u][j] : 0.0f;               // :246-268
                const float c = on ? vb[u][j] : 0.0f;
                s0[j] = q * c;                                      // :420-453
                s1[j] = c * c;
            }
        }
        float r0, r1 = 0.0f;
        if (EUC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s0[j] = s0[j] + quad_xor2(s0[j]);  // i, i+8
#pragma unroll
            for (int j = 0; j < 4; ++j) s0[j] = s0[j] + quad_xor1(s0[j]);  // i, i+4
            r0 = (s0[0] + s0[2]) + (s0[1] + s0[3]);                        // i, i+2 then 0, 1
        } else if (EPL == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0[j] = s0[j] + quad_xor1(s0[j]); s1[j] = s1[j] + quad_xor1(s1[j]); }  // i, i+4
            r0 = (s0[0] + s0[2]) + (s0[1] + s0[3]);
            r1 = (s1[0] + s1[2]) + (s1[1] + s1[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) { s0[j] = s0[j] + quad_xor2(s0[j]); s1[j] = s1[j] + quad_xor2(s1[j]); }  // i, i+4
#pragma unroll
            for (int j = 0; j < 2; ++j) { s0[j] = s0[j] + quad_xor1(s0[j]); s1[j] = s1[j] + quad_xor1(s1[j]); }  // i, i+2
            r0 = s0[0] + s0[1];                                                                                   // 0, 1
            r1 = s1[0] + s1[1];
        }
        if (args.diag == 2) {  // diagnostic: what the loads and the tree sums alone cost
            if (part == 0 && bl < cnt) {
                st<float, NT>(args.out0 + base + bl, r0);
                if (!EUC) st<float, NT>(args.out1 + base + bl, r1);
            }
            continue;
        }
        if (part == 0 && bl < cnt) {
            s_sum0[bl] = r0;
            if (!EUC) s_sum1[bl] = r1;
            s_ctl[bl] = vc[u];
            any_reset |= ((vc[u] >> 16) & 1u) != 0u;
        }
    }
    if (args.diag == 2) return;
    if (args.diag == 1) {  // diagnostic: no accumulation, the beat sums go out as they are
        __syncthreads();
        for (int t0 = tid; t0 < cnt; t0 += kThreads) {
            st<float, NT>(args.out0 + base + t0, s_sum0[t0]);
            if (!EUC) st<float, NT>(args.out1 + base + t0, s_sum1[t0]);
        }
        return;
    }
    // tiles with next to no pair end in them are walked by next to no lane: those fetch four beat sums per LDS round trip
    const bool sparse = __syncthreads_count(any_reset ? 1 : 0) < 4;

    // phase 2: one lane per vector pair accumulates its beats in order; the lane that reaches the end of the tile
    // hands the running sums on (to the next tile, or to the context after the last tile)
    auto hand_on = [&](float acc0, float acc1) {
        if (last_tile) {
            args.state_out[0] = canon_nan(acc0);
            if (!EUC) args.state_out[1] = canon_nan(acc1);
        } else if (args.debug_skip == 0u || blockIdx.x + 1u != args.debug_skip) {
            __hip_atomic_store(slots + 2 * (uint64_t)blockIdx.x, chain_word(args.epoch, acc0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!EUC) __hip_atomic_store(slots + 2 * (uint64_t)blockIdx.x + 1, chain_word(args.epoch, acc1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto walk = [&](int i, float acc0, float acc1) {
        if (!sparse) {
            bool rs;
            do {
                acc0 = acc0 + s_sum0[i];
                s_out0[i] = acc0;
                rs = ((s_ctl[i] >> 16) & 1u) != 0u;
                if (!EUC) {
                    acc1 = acc1 + s_sum1[i];
                    s_out1[i] = acc1;
                }
                ++i;
            } while (i < cnt && !rs);
            if (i == cnt) hand_on(rs ? 0.0f : acc0, rs ? 0.0f : acc1);
        } else {
            bool more = true;
            while (more) {
                float v0[kLook], v1[kLook];
                uint32_t c[kLook];
#pragma unroll
                for (int k = 0; k < kLook; ++k) {
                    v0[k] = s_sum0[i + k];
                    if (!EUC) v1[k] = s_sum1[i + k];
                    c[k] = s_ctl[i + k];
                }
#pragma unroll
                for (int k = 0; k < kLook; ++k) {
                    if (more) {
                        const bool rs = ((c[k] >> 16) & 1u) != 0u;
                        acc0 = acc0 + v0[k];
                        s_out0[i + k] = acc0;
                        if (!EUC) {
                            acc1 = acc1 + v1[k];
                            s_out1[i + k] = acc1;
                        }
                        const bool last = (i + k + 1 == cnt);
                        if (last) hand_on(rs ? 0.0f : acc0, rs ? 0.0f : acc1);
                        more = !rs && !last;
                    }
                }
                i += kLook;
            }
        }
    };
    for (int t0 = tid; t0 < cnt; t0 += kThreads) {
        if (t0 != 0 && (((s_ctl[t0 - 1] >> 16) & 1u) != 0u)) walk(t0, 0.0f, 0.0f);
    }
    if (tid == 0) {  // beat 0 of the tile: the stream's first beat, a fresh pair, or the pair that runs on from the previous tile
        float acc0 = 0.0f, acc1 = 0.0f;
        if (base == 0) {
            acc0 = args.state_in[0];
            if (!EUC) acc1 = args.state_in[1];
        } else if (((before >> 16) & 1u) == 0u) {
            acc0 = chain_wait(slots + 2 * (uint64_t)(blockIdx.x - 1), args.epoch, header, spins);
            if (!EUC) acc1 = chain_wait(slots + 2 * (uint64_t)(blockIdx.x - 1) + 1, args.epoch, header, spins);
        }
        walk(0, acc0, acc1);
    }
    __syncthreads();

    // phase 3: coalesced result store
    for (int t0 = tid; t0 < cnt; t0 += kThreads) {
        st<float, NT>(args.out0 + base + t0, canon_nan(s_out0[t0]));
        if (!EUC) st<float, NT>(args.out1 + base + t0, canon_nan(s_out1[t0]));
    }
}

// ---- the pipelined shape ---------------------------------------------------------------------------------------
// The chained shape as a persistent workgroup whose waves have different jobs, so that its phases overlap instead of
// following each other.  A one-tile workgroup is strictly phase-ordered -- loads, tree sums, walk, store -- and only other
// resident workgroups keep the memory pipeline busy during its walk, which is why the one-tile shapes of the angular beat
// want every wave they can get while their loads would prefer three or four workgroups per CU.  Here
//   waves 0..2 ("loaders")  hold the operand loads of tile t + 2 in flight, and turn tile t + 1 (loaded one round earlier)
//                           into per-beat tree sums in one of two LDS buffers;
//   wave 3     ("walker")   meanwhile accumulates tile t out of the other buffer -- pair starts compacted with ballots, one
//                           lane per vector pair, the lane of beat 0 continuing the pair that runs in from the previous
//                           tile -- and stores its results.
// One workgroup barrier per tile swaps the buffers.  The walker issues no operand loads, so its polls for the predecessor's
// hand-over never wait behind a prefetch (in an earlier form, where all four waves prefetched and then walked, the first
// poll of a wave drained its outstanding loads first and nothing overlapped: 0.64 of the HBM roof).
//
// Workgroup g of G takes tiles g, g + G, g + 2 G, ...; G is at most what the device keeps resident at once (launcher), so
// every workgroup is running when another one waits for its hand-over: tile t - 1 belongs to workgroup g - 1 (or G - 1, one
// round earlier), which publishes after the walk of its tile's last pair -- a walk that waits for nobody unless that one
// pair spans the whole tile.  (A ticket counter that hands tiles out in order would make this independent of residency,
// but one device-scope atomic per tile on a single word runs at about 20 M per second across the XCDs: 1.7 ms for the
// 32768 tiles of 2^24 angular beats, ten times the kernel.)  Hand-over words, epoch, give-up count, the bounded wait
// with its host-side repair, and the arithmetic are those of k_vec_chain, so the results are the same bits.

// Tree sums of one beat part held by this lane: a / b = four elements, mbits = their mask bits (:221-245, :399-419 and the
// adder tree :632-649, :783-803, :915-935, :1007-1027; angular :246-268, :420-453, :650-676, :804-832, :936-964).
template <bool EUC> __device__ __forceinline__ void beat_tree(const f4v &a, const f4v &b, uint32_t mbits, float &r0, float &r1) {
    float s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool on = (mbits >> j) & 1u;
        if (EUC) {
            const float d = on ? (a[j] - b[j]) : 0.0f;
            s0[j] = d * d;
        } else {
            const float q = on ? a[j] : 0.0f;
            const float c = on ? b[j] : 0.0f;
            s0[j] = q * c;
            s1[j] = c * c;
        }
    }
    r1 = 0.0f;
    if (EUC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s0[j] = s0[j] + quad_xor2(s0[j]);  // i, i+8
#pragma unroll
        for (int j = 0; j < 4; ++j) s0[j] = s0[j] + quad_xor1(s0[j]);  // i, i+4
        r0 = (s0[0] + s0[2]) + (s0[1] + s0[3]);                        // i, i+2 then 0, 1
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s0[j] = s0[j] + quad_xor1(s0[j]); s1[j] = s1[j] + quad_xor1(s1[j]); }  // i, i+4
        r0 = (s0[0] + s0[2]) + (s0[1] + s0[3]);
        r1 = (s1[0] + s1[2]) + (s1[1] + s1[3]);
    }
}

constexpr int kPipeLoadWaves = 3;  // of the workgroup's four

// Orders this wave's LDS accesses: what its lanes wrote before is what its lanes read afterwards (no workgroup barrier needed
// inside one wave, whose LDS instructions execute in order).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <bool EUC, bool NT, int PASSES>
__global__ __launch_bounds__(256, 4) void k_vec_pipe(VecArgs args, uint64_t n, uint32_t tiles) {
    constexpr int kLook = 4;                          // beat sums fetched per LDS round trip by the sparse-reset walk
    constexpr int ROW = EUC ? 16 : 8;                 // floats per beat and vector
    constexpr int LPB = ROW / 4;                      // lanes per beat
    constexpr int BPP = kPipeLoadWaves * 64 / LPB;    // beats per pass of the loader waves
    constexpr int SUB = PASSES * BPP;                 // beats per tile

    static_assert(SUB % 4 == 0 && kLook % 4 == 0, "16-byte rows");
    __shared__ float s_sum0[2][SUB + kLook];
    __shared__ float s_sum1[EUC ? 1 : 2][EUC ? 1 : SUB + kLook];
    __shared__ __align__(16) uint32_t s_ctl[2][SUB + kLook];
    __shared__ uint32_t s_before[2];                  // the control word in front of the tile in that buffer
    __shared__ __align__(16) float s_out0[SUB + kLook];
    __shared__ __align__(16) float s_out1[EUC ? 4 : SUB + kLook];
    __shared__ uint16_t s_start[SUB];                 // the walker's list of pair starts

    const int tid = threadIdx.x, lane = tid & 63;
    const bool loader = tid < kPipeLoadWaves * 64;
    const int part = tid % LPB;  // which four elements of the beat
    unsigned long long *const header = args.chain_header;
    unsigned long long *const slots = args.chain;  // two words per tile
    const int spins = args.debug_skip ? kChainSpinsDebug : kChainSpins;
    const uint32_t G = gridDim.x;
    if (blockIdx.x >= tiles) return;
    const bool wide = ((reinterpret_cast<uintptr_t>(args.out0) | (EUC ? 0u : reinterpret_cast<uintptr_t>(args.out1))) & 15u) == 0u;

    f4v va[PASSES], vb[PASSES];
    uint32_t vc[PASSES];
    uint32_t before = 0x10000u;
    auto count_of = [&](uint32_t t) -> int {
        const uint64_t left = n - (uint64_t)t * SUB;
        return (left < (uint64_t)SUB) ? (int)left : SUB;
    };
    auto load = [&](uint32_t t) {  // loaders
        const uint64_t base = (uint64_t)t * SUB;
        const int cnt = count_of(t);
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            const int bl = u * BPP + tid / LPB;
            if (bl < cnt) {
                const uint64_t e = (base + bl) * ROW + part * 4;
                va[u] = ld<f4v, NT>(args.a + e);
                vb[u] = ld<f4v, NT>(args.b + e);
                vc[u] = args.ctl[base + bl];
            } else {
                va[u] = (f4v)(0.0f); vb[u] = (f4v)(0.0f); vc[u] = 0u;
            }
        }
        before = (tid == 0 && base != 0) ? args.ctl[base - 1] : 0x10000u;
    };
    auto sums = [&](uint32_t t, int buf) {  // loaders: the registers hold tile t
        const int cnt = count_of(t);
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            const int bl = u * BPP + tid / LPB;
            float r0, r1;
            beat_tree<EUC>(va[u], vb[u], vc[u] >> (part * 4), r0, r1);
            if (part == 0 && bl < cnt) {
                s_sum0[buf][bl] = r0;
                if (!EUC) s_sum1[buf][bl] = r1;
                s_ctl[buf][bl] = vc[u];
            }
        }
        if (tid == 0) s_before[buf] = before;
    };

    // prologue: the first tile's sums into buffer 0, the second tile's loads in flight
    uint32_t tile = blockIdx.x;
    if (loader) {
        load(tile);
        sums(tile, 0);
        if (tile + G < tiles) load(tile + G);
    }
    __syncthreads();

    int p = 0;
    for (;; tile += G, p ^= 1) {
        const uint32_t next = tile + G;  // (tiles < 2^31 and the grid is small: no wrap)
        if (loader) {
            if (next < tiles) {
                sums(next, p ^ 1);
                if (next + G < tiles) load(next + G);  // in flight for two rounds of the walker
            }
        } else {
            const uint64_t base = (uint64_t)tile * SUB;
            const int cnt = count_of(tile);
            const bool last_tile = (tile + 1u == tiles);
            const float *sum0 = s_sum0[p], *sum1 = s_sum1[EUC ? 0 : p];
            const uint32_t *ctl = s_ctl[p];
            // Does a pair run into this tile?  Then its running sums come from the previous tile's walker: ask for them now,
            // look at the answer after the other pairs have been walked (a round trip to memory under full load takes
            // longer than those walks).
            const bool entering = lane == 0 && tile != 0u && (((s_before[p] >> 16) & 1u) == 0u);
            unsigned long long *const from = slots + 2 * (uint64_t)(tile - (tile ? 1u : 0u));
            unsigned long long got0 = 0ull, got1 = 0ull;
            if (entering) {
                got0 = __hip_atomic_load(from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!EUC) got1 = __hip_atomic_load(from + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // the pair starts of the tile (any order: every pair is walked by a lane of its own); a lane looks at four positions
            int n_starts = 0;
            for (int q0 = 0; 4 * q0 < cnt; q0 += 64) {
                const int t = 4 * (q0 + lane);
                const bool in = t < cnt;
                const u4v c = in ? *reinterpret_cast<const u4v *>(ctl + t) : u4v{0u, 0u, 0u, 0u};
                const uint32_t prev = (in && t != 0) ? ctl[t - 1] : 0u;
                const bool st[4] = {((prev >> 16) & 1u) != 0u, t + 1 < cnt && ((c.x >> 16) & 1u) != 0u, t + 2 < cnt && ((c.y >> 16) & 1u) != 0u,
                                    t + 3 < cnt && ((c.z >> 16) & 1u) != 0u};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long m = __ballot(st[j]);
                    if (st[j]) s_start[n_starts + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(t + j);
                    n_starts += __popcll(m);
                }
            }
            // this wave's walk is the critical path of the round, so every pair is walked in the form that fetches four
            // beat sums per LDS round trip (the padded arrays keep the look-ahead in bounds)
            constexpr bool sparse = true;
            wave_lds_fence();

            auto hand_on = [&](float acc0, float acc1) {
                if (last_tile) {
                    args.state_out[0] = canon_nan(acc0);
                    if (!EUC) args.state_out[1] = canon_nan(acc1);
                } else if (args.debug_skip == 0u || tile + 1u != args.debug_skip) {
                    __hip_atomic_store(slots + 2 * (uint64_t)tile, chain_word(args.epoch, acc0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!EUC) __hip_atomic_store(slots + 2 * (uint64_t)tile + 1, chain_word(args.epoch, acc1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            auto walk = [&](int i, float acc0, float acc1) {
                if (!sparse) {
                    bool rs;
                    do {
                        acc0 = acc0 + sum0[i];
                        s_out0[i] = acc0;
                        rs = ((ctl[i] >> 16) & 1u) != 0u;
                        if (!EUC) {
                            acc1 = acc1 + sum1[i];
                            s_out1[i] = acc1;
                        }
                        ++i;
                    } while (i < cnt && !rs);
                    if (i == cnt) hand_on(rs ? 0.0f : acc0, rs ? 0.0f : acc1);
                } else {
                    bool more = true;
                    while (more) {
                        float v0[kLook], v1[kLook];
                        uint32_t c[kLook];
#pragma unroll
                        for (int k = 0; k < kLook; ++k) {
                            v0[k] = sum0[i + k];
                            if (!EUC) v1[k] = sum1[i + k];
                            c[k] = ctl[i + k];
                        }
#pragma unroll
                        for (int k = 0; k < kLook; ++k) {
                            if (more) {
                                const bool rs = ((c[k] >> 16) & 1u) != 0u;
                                acc0 = acc0 + v0[k];
                                s_out0[i + k] = acc0;
                                if (!EUC) {
                                    acc1 = acc1 + v1[k];
                                    s_out1[i + k] = acc1;
                                }
                                const bool last = (i + k + 1 == cnt);
                                if (last) hand_on(rs ? 0.0f : acc0, rs ? 0.0f : acc1);
                                more = !rs && !last;
                            }
                        }
                        i += kLook;
                    }
                }
            };
            for (int j = lane; j < n_starts; j += 64) walk((int)s_start[j], 0.0f, 0.0f);
            if (lane == 0) {  // beat 0 of the tile: the stream's first beat, a fresh pair, or the pair that runs on from the previous tile
                float acc0 = 0.0f, acc1 = 0.0f;
                if (tile == 0u) {
                    acc0 = args.state_in[0];
                    if (!EUC) acc1 = args.state_in[1];
                } else if (entering) {
                    int spin = 0;
                    while ((uint32_t)(got0 >> 32) != args.epoch || (!EUC && (uint32_t)(got1 >> 32) != args.epoch)) {
                        if (++spin > spins) {  // bounded: counted, repaired by the host (api.cpp)
                            atomicAdd(header, 1ull);
                            got0 = got1 = 0ull;  // float bits of +0
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                        got0 = __hip_atomic_load(from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (!EUC) got1 = __hip_atomic_load(from + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    acc0 = __uint_as_float((uint32_t)got0);
                    if (!EUC) acc1 = __uint_as_float((uint32_t)got1);
                }
                walk(0, acc0, acc1);
            }
            wave_lds_fence();
            // coalesced result store, 16 bytes per lane where the arrays allow it
            if (wide) {
                for (int q0 = 0; 4 * q0 < cnt; q0 += 64) {
                    const int t = 4 * (q0 + lane);
                    if (t + 4 <= cnt) {
                        f4v o0 = *reinterpret_cast<const f4v *>(s_out0 + t);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o0[j] = canon_nan(o0[j]);
                        st<f4v, NT>(args.out0 + base + t, o0);
                        if (!EUC) {
                            f4v o1 = *reinterpret_cast<const f4v *>(s_out1 + t);
#pragma unroll
                            for (int j = 0; j < 4; ++j) o1[j] = canon_nan(o1[j]);
                            st<f4v, NT>(args.out1 + base + t, o1);
                        }
                    } else {
                        for (int i = t; i < cnt; ++i) {
                            st<float, NT>(args.out0 + base + i, canon_nan(s_out0[i]));
                            if (!EUC) st<float, NT>(args.out1 + base + i, canon_nan(s_out1[i]));
                        }
                    }
                }
            } else {
                for (int t0 = lane; t0 < cnt; t0 += 64) {
                    st<float, NT>(args.out0 + base + t0, canon_nan(s_out0[t0]));
                    if (!EUC) st<float, NT>(args.out1 + base + t0, canon_nan(s_out1[t0]));
                }
            }
            wave_lds_fence();  // s_out and s_start are re-used in the next round
        }
        if (next >= tiles) break;
        __syncthreads();  // buffer p ^ 1 holds the next tile's sums; buffer p is free for the one after it
    }
}

size_t vec_chain_bytes(uint64_t n) { return 256 + ((n + 127) / 128) * 16; }  // (256 bytes of slack behind the last tile)

// ------------------------------------------------------- ray pre-processing
// The step in front of the datapath: the ray fields that the reference's software model
// derives from (origin, dir) before a record is poked into the RTL (reference
// src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:36-82):
//   inv = 1.0f / dir                         (IEEE-754 correctly rounded division)
//   kz  = index of the largest |dir| component, ties keep the lower axis (strict >, :36-53)
//   kx  = (kz + 1) % 3, ky = (kx + 1) % 3, exchanged when dir[kz] < 0     (:64-76)
//   shear = (dir[kx] / dir[kz], dir[ky] / dir[kz], 1.0f / dir[kz])         (:78-82)
//   extent = 4000.0f                                                        (:55, :91-92)
// NaN results are written as 0x7FC00000; every consumer in the datapath is blind to
// NaN sign and payload.
__global__ __launch_bounds__(256) void k_ray_prepare(RayPrepArgs a, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float d[3] = {a.dir[0][i], a.dir[1][i], a.dir[2][i]};
    float inv[3], shear[3];
    uint32_t k;
    ray_derive(d, inv, k, shear);
    if (a.inv[0]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.inv[c][i] = inv[c];
    }
    if (a.k) a.k[i] = k;
    if (a.shear[0]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.shear[c][i] = shear[c];
    }
    if (a.extent) a.extent[i] = 4000.0f;
}

hipError_t launch_ray_prepare(const RayPrepArgs &args, uint64_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_ray_prepare, dim3((unsigned)blocks), dim3(256), 0, stream, args, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------- launchers

static inline int min_align(const void *const *ptrs, int count) {
    uintptr_t acc = 0;
    for (int i = 0; i < count; ++i) acc |= reinterpret_cast<uintptr_t>(ptrs[i]);
    if ((acc & 15u) == 0) return 16;
    if ((acc & 7u) == 0) return 8;
    return 4;
}

// Residency cap.  These kernels stream: once the bytes in flight cover latency x bandwidth, more resident waves
// only lengthen the queues in front of the memory channels and scatter the DRAM page hits (measured: the box
// kernel at 2^24 ops runs at 0.765 of the HBM peak with the 7 workgroups per CU its registers allow, 0.82 with
// 3).  A launch states how many 256-thread workgroups a CU may hold by asking for that share of the CU's 160 KiB
// of LDS as dynamic shared memory, which the kernel never touches.
constexpr int kBoxWgsPerCu = 3, kTriWgsPerCu = 5, kEucWgsPerCu = 4, kVecWgsPerCu = 99;  // from sweeps on MI355X
constexpr int kTiledWgsPerCu = 3;  // box and triangle kernels over the tiled layout

static unsigned residency_lds_guess(int w, unsigned static_lds) {
    // gfx950 hands out LDS in granules of 1280 bytes, 128 of them per CU (measured: the plateaus of the
    // kernel time against the requested size end at 20480, 23040, 26880, 32000, 40960 and 53760 bytes)
    const unsigned share = (128u / (unsigned)w) * 1280u;  // w shares fit, w + 1 do not
    return share > static_lds ? share - static_lds : 0u;
}

// The granule above is a measurement, not an architectural promise, so the size is verified with the runtime's
// occupancy query once per (device, kernel, residency): if the guess does not yield exactly `w` resident workgroups,
// sizes between the share of w + 1 and the share of w workgroups are tried until one does; failing that, the measured
// size stays in use.  `achieved` is the query's answer for the size in use, so a caller can see a cap that did not take.
struct Residency { unsigned lds; int achieved; };

constexpr unsigned kMaxLdsPerWorkgroup = 65536u;  // static + dynamic; more would need a function attribute

static Residency resolve_residency(const void *kernel, int w, unsigned static_lds, int threads = 256) {
    auto occupancy = [&](unsigned lds) -> int {
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, threads, lds) != hipSuccess) { (void)hipGetLastError(); return -1; }
        return blocks;
    };
    const unsigned room = kMaxLdsPerWorkgroup > static_lds ? kMaxLdsPerWorkgroup - static_lds : 0u;
    unsigned guess = residency_lds_guess(w, static_lds);
    if (guess > room) guess = room;  // the kernel's own LDS leaves less than the share: the launch must stay launchable
    int got = occupancy(guess);
    if (got == w || got < 0) return {guess, got < 0 ? w : got};  // query unavailable: keep the measured size
    const unsigned total = 160u * 1024u;
    unsigned lo = total / (unsigned)(w + 1), hi = total / (unsigned)w;
    if (hi > kMaxLdsPerWorkgroup) hi = kMaxLdsPerWorkgroup;
    for (unsigned size = hi; size + 1 > lo && size >= 256u; size -= 256u) {
        const unsigned lds = size > static_lds ? size - static_lds : 0u;
        if (lds <= room && occupancy(lds) == w) return {lds, w};
    }
    return {guess, got};  // nothing in the window the hardware's 160 KiB allow: keep the measured size, report the query's answer
}

// `kernel` is the instantiation that is about to be launched: its own static LDS (hipFuncGetAttributes) counts against
// the share, and the answer is cached per device, kernel and residency.
static Residency residency(const void *kernel, int wgs, int dflt) {
    int w = (wgs > 0) ? wgs : dflt;
    if (w >= 99 || w <= 0 || w > 16) return {0u, 0};  // no cap
    if (w < 3) w = 3;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    static std::mutex lock;
    static std::map<std::tuple<int, const void *, int>, Residency> cache;  // filled on first use, read-mostly afterwards
    std::lock_guard<std::mutex> hold(lock);
    const auto key = std::make_tuple(dev, kernel, w);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    hipFuncAttributes attr;
    unsigned static_lds = 0u;
    if (hipFuncGetAttributes(&attr, kernel) == hipSuccess) static_lds = (unsigned)attr.sharedSizeBytes;
    else (void)hipGetLastError();
    const Residency r = resolve_residency(kernel, w, static_lds);
    cache.emplace(key, r);
    return r;
}

static unsigned residency_lds(const void *kernel, int wgs, int dflt) { return residency(kernel, wgs, dflt).lds; }

// The same for a kernel of another file with workgroups of `threads` threads: the dynamic LDS to launch it with so that `w`
// workgroups stay resident per CU (resolved once per device and kernel).
unsigned residency_pad(const void *kernel, int threads, int w, int *achieved) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    static std::mutex lock;
    static std::map<std::tuple<int, const void *, int>, Residency> cache;
    std::lock_guard<std::mutex> hold(lock);
    const auto key = std::make_tuple(dev, kernel, w);
    auto it = cache.find(key);
    if (it == cache.end()) {
        hipFuncAttributes attr;
        unsigned static_lds = 0u;
        if (hipFuncGetAttributes(&attr, kernel) == hipSuccess) static_lds = (unsigned)attr.sharedSizeBytes;
        else (void)hipGetLastError();
        it = cache.emplace(key, resolve_residency(kernel, w, static_lds, threads)).first;
    }
    if (achieved) *achieved = it->second.achieved;
    return it->second.lds;
}

// What the box / triangle launches of a large batch get with these settings: resident 256-thread workgroups per CU
// (0 = no cap) as the occupancy query reports it for the dynamic-LDS size in use.
int resolved_residency(const Tuning &t, bool box) {
    if (box) return residency(reinterpret_cast<const void *>(&k_box<1, true>), t.wgs_per_cu, kBoxWgsPerCu).achieved;
    return residency(reinterpret_cast<const void *>(&k_tri<1, true>), t.wgs_per_cu, kTriWgsPerCu).achieved;
}

#define RTDP_LAUNCH_OPS(KERNEL, ARGS, FIRST, N, V, NT, STREAM, WGS, DFLT)                                      \
    do {                                                                                              \
        const uint64_t groups_ = (N) / (V);                                                           \
        if (groups_) {                                                                                \
            const uint64_t blocks_ = (groups_ + 255) / 256;                                           \
            const unsigned lds_ = residency_lds(reinterpret_cast<const void *>(&KERNEL<V, NT>), WGS, DFLT); \
            hipLaunchKernelGGL((KERNEL<V, NT>), dim3((unsigned)blocks_), dim3(256), lds_, STREAM, ARGS, \
                               (uint64_t)(FIRST), groups_);                                           \
        }                                                                                             \
        const uint64_t tail_ = (N) - groups_ * (V);                                                   \
        if ((V) > 1 && tail_) {                                                                       \
            hipLaunchKernelGGL((KERNEL<1, NT>), dim3(1), dim3(256), 0, STREAM, ARGS, (FIRST) + groups_ * (V), \
                               tail_);                                                                \
        }                                                                                             \
    } while (0)

template <bool NT> static void launch_box_v(const BoxArgs &a, uint64_t first, uint64_t n, int v, int wgs, int dflt, hipStream_t s) {
    if (v == 4) RTDP_LAUNCH_OPS(k_box, a, first, n, 4, NT, s, wgs, dflt);
    else if (v == 2) RTDP_LAUNCH_OPS(k_box, a, first, n, 2, NT, s, wgs, dflt);
    else RTDP_LAUNCH_OPS(k_box, a, first, n, 1, NT, s, wgs, dflt);
}
template <bool NT> static void launch_tri_v(const TriArgs &a, uint64_t first, uint64_t n, int v, int wgs, int dflt, hipStream_t s) {
    if (v == 4) RTDP_LAUNCH_OPS(k_tri, a, first, n, 4, NT, s, wgs, dflt);
    else if (v == 2) RTDP_LAUNCH_OPS(k_tri, a, first, n, 2, NT, s, wgs, dflt);
    else RTDP_LAUNCH_OPS(k_tri, a, first, n, 1, NT, s, wgs, dflt);
}
template <bool NT> static void launch_box_derive_v(const BoxArgs &a, uint64_t first, uint64_t n, int v, int wgs, int dflt, hipStream_t s) {
    if (v == 4) RTDP_LAUNCH_OPS(k_box_derive, a, first, n, 4, NT, s, wgs, dflt);
    else if (v == 2) RTDP_LAUNCH_OPS(k_box_derive, a, first, n, 2, NT, s, wgs, dflt);
    else RTDP_LAUNCH_OPS(k_box_derive, a, first, n, 1, NT, s, wgs, dflt);
}
template <bool NT> static void launch_tri_derive_v(const TriArgs &a, uint64_t first, uint64_t n, int v, int wgs, int dflt, hipStream_t s) {
    if (v == 4) RTDP_LAUNCH_OPS(k_tri_derive, a, first, n, 4, NT, s, wgs, dflt);
    else if (v == 2) RTDP_LAUNCH_OPS(k_tri_derive, a, first, n, 2, NT, s, wgs, dflt);
    else RTDP_LAUNCH_OPS(k_tri_derive, a, first, n, 1, NT, s, wgs, dflt);
}

// Operand and result streams are touched exactly once, so by default they carry the
// nt hint and do not displace each other in L2 / Infinity Cache.
static bool use_nt(const Tuning &t) { return t.nontemporal != 2; }

static int pick_v(int requested, int align_bytes, int dflt) {
    int v = requested > 0 ? requested : dflt;
    if (v != 1 && v != 2 && v != 4) v = dflt;
    while (v * 4 > align_bytes) v >>= 1;
    return v < 1 ? 1 : v;
}

hipError_t launch_box(const BoxArgs &args, uint64_t n, const Tuning &t, hipStream_t stream, bool derive) {
    if (n == 0) return hipSuccess;
    if (t.variant == 10 && !derive) {  // diagnostic: memory pattern only, 8-byte-aligned even-sized batches
        const uint64_t groups = n / 2;
        hipLaunchKernelGGL((k_box_pattern<2>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream, args, groups);
        return hipGetLastError();
    }
    const void *ptrs[kBoxFields + 5];
    for (int i = 0; i < kBoxFields; ++i) ptrs[i] = (derive && i >= 6 && i <= 8) ? nullptr : args.in[i];
    for (int i = 0; i < 4; ++i) ptrs[kBoxFields + i] = args.tmin[i];
    ptrs[kBoxFields + 4] = args.flags;
    const int v = pick_v(t.ops_per_lane, min_align(ptrs, kBoxFields + 5), 1);  // measured with the residency cap: 1 beats 2 and 4 from 2^20 to 2^25 ops
    const int dflt = n >= (uint64_t(1) << 20) ? kBoxWgsPerCu : 99;
    if (derive) {
        if (use_nt(t)) launch_box_derive_v<true>(args, 0, n, v, t.wgs_per_cu, dflt, stream);
        else launch_box_derive_v<false>(args, 0, n, v, t.wgs_per_cu, dflt, stream);
    } else if (use_nt(t)) launch_box_v<true>(args, 0, n, v, t.wgs_per_cu, dflt, stream);
    else launch_box_v<false>(args, 0, n, v, t.wgs_per_cu, dflt, stream);
    return hipGetLastError();
}

static bool grid_fits(uint64_t n) { return (n + 255) / 256 <= 0x7FFFFFFFull; }

hipError_t launch_box_tiled(const float *in, uint32_t *out, uint64_t n, const Tuning &t, hipStream_t stream, bool derive) {
    if (n == 0) return hipSuccess;
    if (!grid_fits(n)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    // measured (2^17 .. 2^26 ops): below 2^20 ops every wave that fits, 2^20 .. 2^21 five workgroups per CU (0.82 against 0.78 with three), from 2^21 three
    const int dflt = n >= (uint64_t(1) << 21) ? kTiledWgsPerCu : (n >= (uint64_t(1) << 20) ? 5 : 99);
    if (derive) {
        if (use_nt(t)) hipLaunchKernelGGL((k_box_tiled<true, true>), grid, block, residency_lds(reinterpret_cast<const void *>(&k_box_tiled<true, true>), t.wgs_per_cu, dflt), stream, in, out, n);
        else hipLaunchKernelGGL((k_box_tiled<false, true>), grid, block, residency_lds(reinterpret_cast<const void *>(&k_box_tiled<false, true>), t.wgs_per_cu, dflt), stream, in, out, n);
    } else {
        if (use_nt(t)) hipLaunchKernelGGL((k_box_tiled<true, false>), grid, block, residency_lds(reinterpret_cast<const void *>(&k_box_tiled<true, false>), t.wgs_per_cu, dflt), stream, in, out, n);
        else hipLaunchKernelGGL((k_box_tiled<false, false>), grid, block, residency_lds(reinterpret_cast<const void *>(&k_box_tiled<false, false>), t.wgs_per_cu, dflt), stream, in, out, n);
    }
    return hipGetLastError();
}

hipError_t launch_tri_tiled(const float *in, uint32_t *out, uint64_t n, const Tuning &t, hipStream_t stream, bool derive) {
    if (n == 0) return hipSuccess;
    if (!grid_fits(n)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    // measured: below 2^21 ops uncapped (2^20: 0.91 against 0.80 with three workgroups per CU), 2^21 .. 2^22 five (0.84 against 0.77), from 2^22 three
    const int dflt = n >= (uint64_t(1) << 22) ? kTiledWgsPerCu : (n >= (uint64_t(1) << 21) ? 5 : 99);
    if (derive) {
        if (use_nt(t)) hipLaunchKernelGGL((k_tri_tiled<true, true>), grid, block, residency_lds(reinterpret_cast<const void *>(&k_tri_tiled<true, true>), t.wgs_per_cu, dflt), stream, in, out, n);
        else hipLaunchKernelGGL((k_tri_tiled<false, true>), grid, block, residency_lds(reinterpret_cast<const void *>(&k_tri_tiled<false, true>), t.wgs_per_cu, dflt), stream, in, out, n);
    } else {
        if (use_nt(t)) hipLaunchKernelGGL((k_tri_tiled<true, false>), grid, block, residency_lds(reinterpret_cast<const void *>(&k_tri_tiled<true, false>), t.wgs_per_cu, dflt), stream, in, out, n);
        else hipLaunchKernelGGL((k_tri_tiled<false, false>), grid, block, residency_lds(reinterpret_cast<const void *>(&k_tri_tiled<false, false>), t.wgs_per_cu, dflt), stream, in, out, n);
    }
    return hipGetLastError();
}

static bool tile_copy_aligned(const TileList &arrays, int count, const void *tiled) {
    uintptr_t acc = reinterpret_cast<uintptr_t>(tiled);
    for (int k = 0; k < count; ++k) acc |= reinterpret_cast<uintptr_t>(arrays.a[k]);
    return (acc & 15u) == 0;
}

hipError_t launch_tile_pack(const TileList &arrays, int count, uint64_t n, uint32_t *tiled, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (count < 1 || count > kTileMaxArrays || !grid_fits(n)) return hipErrorInvalidValue;
    const uint64_t runs = ((n + kTileOps - 1) / kTileOps) * (uint64_t)count;
    if (tile_copy_aligned(arrays, count, tiled) && runs * kCopySplit <= 0x7FFFFFFFull)
        hipLaunchKernelGGL((k_tile_copy16<true>), dim3((unsigned)(runs * kCopySplit)), dim3(256),
                           residency_lds(reinterpret_cast<const void *>(&k_tile_copy16<true>), 0, kCopyWgsPerCu), stream, arrays, (uint32_t)count, n, tiled);
    else
        hipLaunchKernelGGL(k_tile_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, arrays, count, n, tiled);
    return hipGetLastError();
}

hipError_t launch_tile_unpack(const uint32_t *tiled, int count, uint64_t n, const TileList &arrays, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (count < 1 || count > kTileMaxArrays || !grid_fits(n)) return hipErrorInvalidValue;
    const uint64_t runs = ((n + kTileOps - 1) / kTileOps) * (uint64_t)count;
    if (tile_copy_aligned(arrays, count, tiled) && runs * kCopySplit <= 0x7FFFFFFFull)
        hipLaunchKernelGGL((k_tile_copy16<false>), dim3((unsigned)(runs * kCopySplit)), dim3(256),
                           residency_lds(reinterpret_cast<const void *>(&k_tile_copy16<false>), 0, kCopyWgsPerCu), stream, arrays, (uint32_t)count, n, const_cast<uint32_t *>(tiled));
    else
        hipLaunchKernelGGL(k_tile_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tiled, count, n, arrays);
    return hipGetLastError();
}

hipError_t launch_tri(const TriArgs &args, uint64_t n, const Tuning &t, hipStream_t stream, bool derive) {
    if (n == 0) return hipSuccess;
    const void *ptrs[kTriFields + 4];
    for (int i = 0; i < kTriFields; ++i) ptrs[i] = args.in[i];
    ptrs[kTriFields] = derive ? nullptr : args.k;
    ptrs[kTriFields + 1] = args.t_num;
    ptrs[kTriFields + 2] = args.t_denom;
    ptrs[kTriFields + 3] = args.hit;
    const int v = pick_v(t.ops_per_lane, min_align(ptrs, kTriFields + 4), 1);  // measured: never slower than 2, 5-20 % faster below 2^24 ops
    const int dflt = n >= (uint64_t(1) << 21) ? kTriWgsPerCu : 99;
    if (derive) {
        if (use_nt(t)) launch_tri_derive_v<true>(args, 0, n, v, t.wgs_per_cu, dflt, stream);
        else launch_tri_derive_v<false>(args, 0, n, v, t.wgs_per_cu, dflt, stream);
    } else if (use_nt(t)) launch_tri_v<true>(args, 0, n, v, t.wgs_per_cu, dflt, stream);
    else launch_tri_v<false>(args, 0, n, v, t.wgs_per_cu, dflt, stream);
    return hipGetLastError();
}

// The residency cap of a beat launch is resolved for the instantiation that is launched (its static LDS differs with PASSES).
#define RTDP_LAUNCH_VEC(KERNEL, BLOCKS, WGS, DFLT, STREAM, ...)                                                         \
    hipLaunchKernelGGL((KERNEL), dim3(BLOCKS), dim3(kThreads), residency_lds(reinterpret_cast<const void *>(&KERNEL), WGS, DFLT), STREAM, __VA_ARGS__)

template <bool EUC, bool NT> static void launch_vec_p(const VecArgs &args, uint64_t n, uint32_t tile, int passes, int walk, unsigned blocks, int wgs, int dflt,
                                                      hipStream_t stream) {
    switch (passes) {
    case 2: RTDP_LAUNCH_VEC((k_vec<EUC, NT, 2>), blocks, wgs, dflt, stream, args, n, tile, walk); break;
    case 8: RTDP_LAUNCH_VEC((k_vec<EUC, NT, 8>), blocks, wgs, dflt, stream, args, n, tile, walk); break;
    default: RTDP_LAUNCH_VEC((k_vec<EUC, NT, 4>), blocks, wgs, dflt, stream, args, n, tile, walk); break;
    }
}

template <bool EUC, bool NT> static void launch_chain_p(const VecArgs &args, uint64_t n, int passes, int wgs, int dflt, hipStream_t stream) {
    const uint64_t sub = (uint64_t)passes * (kThreads / (EUC ? 4 : 2));
    const unsigned blocks = (unsigned)((n + sub - 1) / sub);
    switch (passes) {
    case 2: RTDP_LAUNCH_VEC((k_vec_chain<EUC, NT, 2, 4>), blocks, wgs, dflt, stream, args, n); break;
    case 8: RTDP_LAUNCH_VEC((k_vec_chain<EUC, NT, 8, 4>), blocks, wgs, dflt, stream, args, n); break;
    default: RTDP_LAUNCH_VEC((k_vec_chain<EUC, NT, 4, 4>), blocks, wgs, dflt, stream, args, n); break;
    }
}

static int cu_count() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        return 256;
    }
    return cus;
}

// The pipelined shape is a persistent grid: as many workgroups as stay resident together (residency x CUs), fewer when
// the segment has fewer tiles; the residency cap is applied like everywhere else, through dynamic LDS.
template <bool EUC, bool NT> static void launch_pipe_p(const VecArgs &args, uint64_t n, uint32_t tiles, int passes, int wgs, int dflt, hipStream_t stream) {
    int w = (wgs > 0 && wgs <= 16) ? wgs : dflt;
    if (w < 3) w = 3;
    const uint64_t resident = (uint64_t)w * (uint64_t)cu_count();
    const unsigned blocks = (unsigned)(tiles < resident ? tiles : resident);
    switch (passes) {
    case 2: RTDP_LAUNCH_VEC((k_vec_pipe<EUC, NT, 2>), blocks, w, w, stream, args, n, tiles); break;
    case 8: RTDP_LAUNCH_VEC((k_vec_pipe<EUC, NT, 8>), blocks, w, w, stream, args, n, tiles); break;
    default: RTDP_LAUNCH_VEC((k_vec_pipe<EUC, NT, 4>), blocks, w, w, stream, args, n, tiles); break;
    }
}

// Which beat kernel a segment of n beats gets under tuning t (chain_ok: a hand-over workspace is available, i.e. the
// launch is not being captured into a graph).  One place decides, the launcher and kernel_name_* both read it.
struct VecShape { bool chained; bool pipe; int passes; uint32_t tile; int walk; int dflt_wgs; uint32_t diag; };

template <bool EUC> static VecShape vec_shape(uint64_t n, const Tuning &t, bool chain_ok) {
    // Two shapes with identical results.  Chained: fixed tiles that hand their running sums on (needs the workspace).
    // Owning: a workgroup owns whole vector pairs and searches for their boundaries.  Measured inside the benchmark
    // process (2^24 beats): Euclidean 0.814 chained against 0.800 owning, angular 0.745 chained against 0.762 owning,
    // so each opcode defaults to its better shape.  Variant 23 (21, 22): owning for both; 27 (24, 25): chained for both.
    // Short angular segments (the quarters of a mixed batch, the segments behind the opcode partition of a record stream) are
    // the exception: up to 2^23 beats the chained shape leads on every box measured (2^22 beats: 0.66 against 0.64 of the HBM
    // peak, 2^20: 0.52 against 0.49), above that the two trade places by less than the box-to-box spread.
    VecShape sh{};
    const bool short_angular = !EUC && n <= (1ull << 23);
    sh.chained = chain_ok && (t.variant == 27 || t.variant == 24 || t.variant == 25 || ((EUC || short_angular) && !(t.variant >= 21 && t.variant <= 23)));
    const bool asked = (t.vec_passes == 2 || t.vec_passes == 4 || t.vec_passes == 8);
    if (chain_ok && t.variant == 28) {  // the pipelined persistent shape (same workspace as the chained one); measured slower, never a default
        sh.chained = sh.pipe = true;
        sh.passes = asked ? t.vec_passes : (EUC ? 8 : 4);
        if (EUC && sh.passes < 4) sh.passes = 4;
        sh.dflt_wgs = 4;
        return sh;
    }
    if (sh.chained) {
        sh.diag = (t.variant == 24) ? 1u : (t.variant == 25) ? 2u : 0u;
        // Euclidean: eight load pairs per lane on long segments; on short ones twice as many, half as long workgroups fill the
        // chip better (4 against 8: 2^18 beats 0.44 / 0.40 of the HBM peak, 2^20 0.66 / 0.62, 2^22 equal, 2^24 0.81 / 0.83)
        sh.passes = asked ? t.vec_passes : (EUC ? (n <= (1ull << 21) ? 4 : 8) : 4);
        // residency: the Euclidean shape (64 KiB of loads in flight per workgroup) is best with 4 workgroups per CU (0.824
        // of the HBM roof against 0.79 with the 6 its registers allow), the angular shape with all it can get
        sh.dflt_wgs = (EUC && sh.passes == 8) ? kEucWgsPerCu : kVecWgsPerCu;
        return sh;
    }
    // defaults from interleaved sweeps on MI355X: sub-tiles of 512 beats for both opcodes; the Euclidean shape (loads
    // issued from the nominal start) wants the nominal tile equal to the sub-tile, the angular shape (loads from the
    // first pair boundary) a tile a few beats shorter, so that the pair that hangs over the nominal end still fits
    // the one sub-tile (504: 0.749 of the HBM roof against 0.738 with 512)
    sh.passes = asked ? t.vec_passes : (EUC ? 8 : 4);
    sh.tile = t.vec_tile > 0 ? (uint32_t)t.vec_tile : (EUC ? 512u : 504u);
    if (sh.tile < 64u) sh.tile = 64u;
    sh.walk = (t.variant == 21) ? 1 : (t.variant == 22) ? 2 : 0;  // A/B only: 21 plain walk everywhere, 22 look-ahead walk everywhere
    sh.dflt_wgs = kVecWgsPerCu;
    return sh;
}

template <bool EUC> static hipError_t launch_vec(const VecArgs &args, uint64_t n, const Tuning &t, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const VecShape sh = vec_shape<EUC>(n, t, args.chain != nullptr);
    if (sh.pipe) {
        const uint64_t sub = (uint64_t)sh.passes * (kPipeLoadWaves * 64 / (EUC ? 4 : 2));  // >= 128 beats: the hand-over words are sized for that
        const uint64_t tiles = (n + sub - 1) / sub;
        if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
        if (use_nt(t)) launch_pipe_p<EUC, true>(args, n, (uint32_t)tiles, sh.passes, t.wgs_per_cu, sh.dflt_wgs, stream);
        else launch_pipe_p<EUC, false>(args, n, (uint32_t)tiles, sh.passes, t.wgs_per_cu, sh.dflt_wgs, stream);
        return hipGetLastError();
    }
    if (sh.chained) {
        VecArgs a2 = args;
        a2.diag = sh.diag;
        if ((n + 127) / 128 > 0x7FFFFFFFull) return hipErrorInvalidValue;
        if (use_nt(t)) launch_chain_p<EUC, true>(a2, n, sh.passes, t.wgs_per_cu, sh.dflt_wgs, stream);
        else launch_chain_p<EUC, false>(a2, n, sh.passes, t.wgs_per_cu, sh.dflt_wgs, stream);
        return hipGetLastError();
    }
    const uint64_t blocks = (n + sh.tile - 1) / sh.tile;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (use_nt(t)) launch_vec_p<EUC, true>(args, n, sh.tile, sh.passes, sh.walk, (unsigned)blocks, t.wgs_per_cu, sh.dflt_wgs, stream);
    else launch_vec_p<EUC, false>(args, n, sh.tile, sh.passes, sh.walk, (unsigned)blocks, t.wgs_per_cu, sh.dflt_wgs, stream);
    return hipGetLastError();
}

// The kernels behind a launch, named as a profiler prints them (rocprofv3 --kernel-trace): what a committed counter pass
// is keyed by, so a harness can tell whether the pass still describes the kernel the library picks.
static const char *tf(bool b) { return b ? "true" : "false"; }

void kernel_name_vec(bool euclid, uint64_t n, const Tuning &t, bool chain_ok, char *buf, size_t len) {
    const VecShape sh = euclid ? vec_shape<true>(n, t, chain_ok) : vec_shape<false>(n, t, chain_ok);
    if (sh.pipe) snprintf(buf, len, "rtdp::k_vec_pipe<%s, %s, %d>", tf(euclid), tf(use_nt(t)), sh.passes);
    else if (sh.chained) snprintf(buf, len, "rtdp::k_vec_chain<%s, %s, %d, 4>", tf(euclid), tf(use_nt(t)), sh.passes);
    else snprintf(buf, len, "rtdp::k_vec<%s, %s, %d>", tf(euclid), tf(use_nt(t)), sh.passes);
}

void kernel_name_ops(bool box, bool tiled, bool derive, int ops_per_lane_possible, const Tuning &t, char *buf, size_t len) {
    if (tiled) { snprintf(buf, len, "rtdp::k_%s_tiled<%s, %s>", box ? "box" : "tri", tf(use_nt(t)), tf(derive)); return; }
    if (box && t.variant == 10 && !derive) { snprintf(buf, len, "rtdp::k_box_pattern<2>"); return; }
    snprintf(buf, len, "rtdp::k_%s%s<%d, %s>", box ? "box" : "tri", derive ? "_derive" : "", pick_v(t.ops_per_lane, ops_per_lane_possible * 4, 1), tf(use_nt(t)));
}

hipError_t launch_euclid(const VecArgs &args, uint64_t n, const Tuning &t, hipStream_t stream) { return launch_vec<true>(args, n, t, stream); }
hipError_t launch_angular(const VecArgs &args, uint64_t n, const Tuning &t, hipStream_t stream) { return launch_vec<false>(args, n, t, stream); }

}  // namespace rtdp
