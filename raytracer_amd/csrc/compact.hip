// The consumer step behind a Quadbox batch: turn the sorted hit masks of n ops into one dense list of
// (item, child) visits, closest child first within an op, ops in batch order -- what a traversal loop pushes
// next (no counterpart in the reference's sources; the order it relies on is the Quadbox output order,
// src/main/scala/raytracer_datapath/UnifiedDatapath.scala:1117-1132: hits first, nearest first).
//
//   k_compact_count   hits per 256-op workgroup (popcount of the isIntersect nibbles)
//   k_compact_scan    exclusive scan of those counts (one workgroup)
//   k_compact_emit    every lane knows its offset from ballots: since the hits of an op occupy its first
//                     slots, "lanes with more than j hits" (one ballot per j) counted below the lane give the
//                     lane's offset inside the wave; waves add up through LDS, workgroups through the scan
// The list is a pure function of the inputs (no atomics decide the order).
#include "kernels.h"

#include <mutex>

namespace rtdp {

constexpr int kCmpBlk = 256;

// Where the result words of op i sit: plain arrays (tile_words 0), or a tiled result buffer of tile_words words per op
// whose word k is addressed through a pointer to that word's run in tile 0.
__device__ __forceinline__ uint64_t result_index(uint64_t i, uint32_t tile_words) {
    return tile_words ? (i / kTileOps) * (uint64_t)tile_words * kTileOps + (i % kTileOps) : i;
}

__global__ __launch_bounds__(kCmpBlk) void k_compact_count(const uint32_t *__restrict__ flags, uint64_t n, uint32_t *__restrict__ blockcnt, uint32_t tile_words) {
    __shared__ uint32_t s_w[kCmpBlk / 64];
    const uint64_t i = (uint64_t)blockIdx.x * kCmpBlk + threadIdx.x;
    uint32_t c = (i < n) ? (uint32_t)__popc(flags[result_index(i, tile_words)] & 15u) : 0u;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// counts[nblk] become exclusive offsets in place; total[0..1] receives the 64-bit sum
__global__ __launch_bounds__(1024) void k_compact_scan(uint32_t *__restrict__ counts, uint32_t nblk, unsigned long long *__restrict__ total,
                                                      unsigned long long *__restrict__ offsets64) {
    __shared__ unsigned long long s_part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nblk + 1023u) / 1024u;
    const uint32_t b0 = t * per, b1 = (b0 + per < nblk) ? b0 + per : nblk;
    unsigned long long sum = 0;
    for (uint32_t b = b0; b < b1; ++b) sum += counts[b];
    s_part[t] = sum;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0;
        for (int k = 0; k < 1024; ++k) { const unsigned long long v = s_part[k]; s_part[k] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = s_part[t];
    for (uint32_t b = b0; b < b1; ++b) { offsets64[b] = run; run += counts[b]; }
}

__global__ __launch_bounds__(kCmpBlk) void k_compact_emit(const uint32_t *__restrict__ flags, const float *__restrict__ t0, const float *__restrict__ t1,
                                                         const float *__restrict__ t2, const float *__restrict__ t3, uint64_t n,
                                                         const uint32_t *__restrict__ item, const uint32_t *__restrict__ children,
                                                         const unsigned long long *__restrict__ offsets64, uint64_t capacity,
                                                         uint32_t *__restrict__ out_item, uint32_t *__restrict__ out_child,
                                                         float *__restrict__ out_tmin, uint32_t tile_words) {
    __shared__ uint32_t s_w[kCmpBlk / 64];
    const uint64_t i = (uint64_t)blockIdx.x * kCmpBlk + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t ri = result_index(i, tile_words);
    const uint32_t fl = (i < n) ? flags[ri] : 0u;
    const uint32_t cnt = (uint32_t)__popc(fl & 15u);  // hits come first: they sit in slots 0 .. cnt - 1
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const unsigned long long m = __ballot(cnt > j);
        off += (uint32_t)__popcll(m & below);
        tot += (uint32_t)__popcll(m);
    }
    if (lane == 0) s_w[wave] = tot;
    __syncthreads();
    unsigned long long base = offsets64[blockIdx.x] + off;
    for (int w = 0; w < wave; ++w) base += s_w[w];
    if (cnt == 0) return;
    const uint32_t it = item ? item[i] : (uint32_t)i;
    const float t[4] = {out_tmin ? t0[ri] : 0.0f, out_tmin ? t1[ri] : 0.0f, out_tmin ? t2[ri] : 0.0f, out_tmin ? t3[ri] : 0.0f};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        if (k >= cnt) break;
        const unsigned long long pos = base + k;
        if (pos >= capacity) break;
        const uint32_t slot = (fl >> (4 + 2 * k)) & 3u;
        out_item[pos] = it;
        out_child[pos] = children ? children[i * 4 + slot] : slot;
        if (out_tmin) out_tmin[pos] = t[k];
    }
}

hipError_t compact_hits(const uint32_t *flags, const float *const tmin[4], uint64_t n, const uint32_t *item, const uint32_t *children,
                        uint32_t *blockcnt, unsigned long long *offsets64, unsigned long long *total, uint64_t capacity, uint32_t *out_item,
                        uint32_t *out_child, float *out_tmin, hipStream_t s, uint32_t tile_words) {
    const uint64_t nblk = (n + kCmpBlk - 1) / kCmpBlk;
    if (nblk > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_compact_count, dim3((unsigned)nblk), dim3(kCmpBlk), 0, s, flags, n, blockcnt, tile_words);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, s, blockcnt, (uint32_t)nblk, total, offsets64);
    hipLaunchKernelGGL(k_compact_emit, dim3((unsigned)nblk), dim3(kCmpBlk), 0, s, flags, tmin[0], tmin[1], tmin[2], tmin[3], n, item,
                       children, offsets64, capacity, out_item, out_child, out_tmin, tile_words);
    return hipGetLastError();
}

// ---- the same compaction for callers that cannot allocate -----------------------------------------------
// The block counts and offsets live in a file-scope device array instead of a buffer the context grows, so the
// call path contains no allocation at all.  Every GPU holds its own copy of the array at its own address: the
// address is looked up once per device (compact_static_init, with that device current) and kept; host code
// never uses the array's name for anything else.  Nothing in the array has to start at zero: a call writes
// every count and offset it reads.  Calls that overlap on one GPU share the array and race.
constexpr uint64_t kStaticMaxOps = 1ull << 24;
constexpr uint64_t kStaticBlocks = kStaticMaxOps / kCmpBlk;
static __device__ __align__(256) unsigned char scratch[kStaticBlocks * 12];  // [blocks] u64 offsets, then [blocks] u32 counts: 768 KiB

constexpr int kMaxDevices = 64;
static unsigned char *g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_lock;

uint64_t compact_static_max_ops() { return kStaticMaxOps; }

hipError_t compact_static_init(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (g_scratch_ptr[device]) return hipSuccess;
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = static_cast<unsigned char *>(p);
    return hipSuccess;
}

hipError_t compact_hits_static(int device, const uint32_t *flags, const float *const tmin[4], uint64_t n, const uint32_t *item,
                               const uint32_t *children, uint64_t capacity, uint32_t *out_item, uint32_t *out_child, float *out_tmin,
                               unsigned long long *d_count, hipStream_t s) {
    if (device < 0 || device >= kMaxDevices || n > kStaticMaxOps) return hipErrorInvalidValue;
    unsigned char *base;
    {
        std::lock_guard<std::mutex> hold(g_scratch_lock);
        base = g_scratch_ptr[device];
    }
    if (!base) return hipErrorNotInitialized;
    unsigned long long *offsets64 = reinterpret_cast<unsigned long long *>(base);
    uint32_t *blockcnt = reinterpret_cast<uint32_t *>(base + kStaticBlocks * 8);
    return compact_hits(flags, tmin, n, item, children, blockcnt, offsets64, d_count, capacity, out_item, out_child, out_tmin, s, 0u);
}

}  // namespace rtdp
