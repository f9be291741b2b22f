// The k-d index of the k-NN searches built on the GPU (no counterpart in the reference).  Same definition as the
// host builder: 2^D leaves of leaf_size slots, pads (+Inf, index 0xFFFFFFFF) behind the real points; level by level
// every segment is sorted along its widest finite dimension, which leaves the median split in the middle.
// All segments of a level are sorted by ONE radix sort on 64-bit keys (segment << 32 | ordered image of the
// coordinate, NaN last, -0 as +0), which is stable, so ties keep their current order.
//   k_kd_extent   finite bounding box of every segment (ordered-integer atomics, read before each atomic)
//   k_kd_dim      widest dimension per segment (first one on ties)
//   k_kd_keys     the sort keys of a level
//   k_kd_split    split plane of every segment: the first key of its upper half (heap order: node 2^l - 1 + s)
//   k_kd_gather / k_kd_leaf_boxes   the stored points, their indices and the leaf boxes
//   k_kd_home     the leaf a query reaches by descending the split planes (goes right when coordinate >= plane)
#include <hipcub/hipcub.hpp>

#include "kernels.h"

namespace rtdp {

namespace {

constexpr uint32_t kPad = 0xFFFFFFFFu;
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ordered(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

__global__ __launch_bounds__(256) void k_kd_init(uint32_t *__restrict__ perm, uint32_t total, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < total) perm[i] = i < n ? i : kPad;
}

__global__ __launch_bounds__(256) void k_kd_bounds_init(uint32_t *__restrict__ bounds, uint32_t segs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < segs * 32u) bounds[i] = ((i & 31u) < 16u) ? 0xFFFFFFFFu : 0u;  // [seg][0..15] lo, [16..31] hi
}

// A wave whose 64 positions lie in one segment reduces its values with shuffles first and goes to memory once per
// dimension; waves that straddle segments (the short segments of the deep levels) let every lane go by itself.
// Either way a bound is read before the atomic, which is skipped unless the value improves it.
__global__ __launch_bounds__(256) void k_kd_extent(const float *__restrict__ pts, const uint32_t *__restrict__ perm, uint32_t total,
                                                  uint32_t seglen, uint32_t *__restrict__ bounds) {
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    const bool in = pos < total;
    const uint32_t id = in ? perm[pos] : kPad;
    const uint32_t seg = (in ? pos : total - 1u) / seglen;
    const uint32_t seg0 = __shfl(seg, 0);
    const bool uniform = __all(seg == seg0);
    uint32_t *b = bounds + (size_t)seg * 32;
    const float *p = pts + (size_t)(id == kPad ? 0u : id) * 16;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        if (id != kPad) {
            const float v = p[d];
            if (fabsf(v) <= 3.4028234664e38f) lo = hi = ordered(v);  // NaN and +-Inf do not count
        }
        if (uniform) {
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t l = __shfl_down(lo, off), h = __shfl_down(hi, off);
                lo = l < lo ? l : lo;
                hi = h > hi ? h : hi;
            }
            if ((threadIdx.x & 63u) != 0u) continue;
        }
        if (lo != 0xFFFFFFFFu) {
            if (lo < __hip_atomic_load(&b[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&b[d], lo);
            if (hi > __hip_atomic_load(&b[16 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&b[16 + d], hi);
        }
    }
}

__global__ __launch_bounds__(256) void k_kd_dim(const uint32_t *__restrict__ bounds, uint32_t segs, uint32_t *__restrict__ dim) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= segs) return;
    const uint32_t *b = bounds + (size_t)s * 32;
    uint32_t best_d = 0;
    float best = -1.0f;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        if (b[d] == 0xFFFFFFFFu) continue;  // no finite value in this dimension
        const float ext = unordered(b[16 + d]) - unordered(b[d]);
        if (fabsf(ext) <= 3.4028234664e38f && ext > best) { best = ext; best_d = (uint32_t)d; }
    }
    dim[s] = best_d;
}

__device__ __forceinline__ uint32_t sort_key(float v) {
    if (v != v) return 0xFFFFFFFFu;  // NaN after everything else, the +Inf pads included
    return ordered(v == 0.0f ? 0.0f : v);
}

__global__ __launch_bounds__(256) void k_kd_keys(const float *__restrict__ pts, const uint32_t *__restrict__ perm, uint32_t total, uint32_t seglen,
                                                const uint32_t *__restrict__ dim, unsigned long long *__restrict__ keys) {
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    if (pos >= total) return;
    const uint32_t seg = pos / seglen, id = perm[pos];
    const float v = (id == kPad) ? __uint_as_float(0x7F800000u) : pts[(size_t)id * 16 + dim[seg]];
    keys[pos] = ((unsigned long long)seg << 32) | sort_key(v);
}

__global__ __launch_bounds__(256) void k_kd_split(const unsigned long long *__restrict__ sorted_keys, const uint32_t *__restrict__ dim, uint32_t segs,
                                                 uint32_t seglen, uint32_t *__restrict__ split_dim, float *__restrict__ split_val) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= segs) return;
    const uint32_t node = segs - 1u + s;  // segs = 2^level
    const uint32_t k = (uint32_t)sorted_keys[(size_t)s * seglen + seglen / 2];
    split_dim[node] = dim[s];
    split_val[node] = (k == 0xFFFFFFFFu) ? __uint_as_float(0x7FC00000u) : unordered(k);
}

__global__ __launch_bounds__(256) void k_kd_gather(const float *__restrict__ pts, const uint32_t *__restrict__ perm, uint32_t total, uint32_t index_base,
                                                  float *__restrict__ out_points, uint32_t *__restrict__ out_index) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;  // one thread per 4 coordinates
    const uint32_t pos = g >> 2, q = g & 3u;
    if (pos >= total) return;
    const uint32_t id = perm[pos];
    const float inf = __uint_as_float(0x7F800000u);
    f4v v = {inf, inf, inf, inf};
    if (id != kPad) v = reinterpret_cast<const f4v *>(pts + (size_t)id * 16)[q];
    reinterpret_cast<f4v *>(out_points + (size_t)pos * 16)[q] = v;
    if (q == 0) out_index[pos] = (id == kPad) ? kPad : index_base + id;
}

// one thread per (leaf, dimension); a NaN coordinate in the leaf makes the box unbounded in that dimension
__global__ __launch_bounds__(256) void k_kd_leaf_boxes(const float *__restrict__ out_points, const uint32_t *__restrict__ out_index, uint32_t n_leaves,
                                                      uint32_t leaf_size, float *__restrict__ lo, float *__restrict__ hi) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t leaf = g >> 4, d = g & 15u;
    if (leaf >= n_leaves) return;
    const float inf = __uint_as_float(0x7F800000u);
    float l = inf, h = -inf;
    bool nan = false;
    for (uint32_t s = 0; s < leaf_size; ++s) {
        const size_t pos = (size_t)leaf * leaf_size + s;
        if (out_index[pos] == kPad) continue;
        const float v = out_points[pos * 16 + d];
        nan = nan || (v != v);
        l = (v < l) ? v : l;
        h = (v > h) ? v : h;
    }
    lo[(size_t)leaf * 16 + d] = nan ? -inf : l;
    hi[(size_t)leaf * 16 + d] = nan ? inf : h;
}

__global__ __launch_bounds__(256) void k_kd_home(const float *__restrict__ qry, uint32_t n_qry, uint32_t depth, const uint32_t *__restrict__ split_dim,
                                                const float *__restrict__ split_val, uint32_t *__restrict__ home) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_qry) return;
    const float *q = qry + (size_t)i * 16;
    uint32_t seg = 0;
    for (uint32_t l = 0; l < depth; ++l) {
        const uint32_t node = (1u << l) - 1u + seg;
        const float x = q[split_dim[node] & 15u];
        seg = 2u * seg + ((x >= split_val[node]) ? 1u : 0u);
    }
    home[i] = seg;
}

}  // namespace

size_t kd_device_workspace(uint32_t total, uint32_t n_leaves) {
    size_t temp = 0;
    unsigned long long *kn = nullptr;
    uint32_t *vn = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, kn, kn, vn, vn, (int)total, 0, 64);
    return ((size_t)total * (8 + 8 + 4 + 4) + (size_t)n_leaves * (32 * 4 + 4) + 1024 + temp + 255) & ~size_t(255);
}

hipError_t kd_build_device(const float *pts, uint32_t n, uint32_t n_leaves, uint32_t leaf_size, uint32_t index_base, char *workspace,
                           size_t workspace_bytes, float *out_points, uint32_t *out_index, float *out_lo, float *out_hi, uint32_t *split_dim,
                           float *split_val, hipStream_t s) {
    const uint32_t total = n_leaves * leaf_size;
    unsigned long long *keys[2] = {reinterpret_cast<unsigned long long *>(workspace), reinterpret_cast<unsigned long long *>(workspace) + total};
    uint32_t *perm[2] = {reinterpret_cast<uint32_t *>(keys[1] + total), reinterpret_cast<uint32_t *>(keys[1] + total) + total};
    uint32_t *bounds = perm[1] + total;                    // at most n_leaves / 2 segments x 32 words
    uint32_t *dim = bounds + (size_t)n_leaves * 32;
    char *temp = reinterpret_cast<char *>(dim + n_leaves);
    temp += (256 - (reinterpret_cast<uintptr_t>(temp) & 255)) & 255;
    const size_t temp_bytes = workspace_bytes - (size_t)(temp - workspace);
    const unsigned blocks = (total + 255) / 256;
    uint32_t depth = 0;
    while ((1u << depth) < n_leaves) ++depth;
    hipLaunchKernelGGL(k_kd_init, dim3(blocks), dim3(256), 0, s, perm[0], total, n);
    int cur = 0;
    for (uint32_t level = 0; level < depth; ++level) {
        const uint32_t segs = 1u << level, seglen = total >> level;
        hipLaunchKernelGGL(k_kd_bounds_init, dim3((segs * 32 + 255) / 256), dim3(256), 0, s, bounds, segs);
        hipLaunchKernelGGL(k_kd_extent, dim3(blocks), dim3(256), 0, s, pts, perm[cur], total, seglen, bounds);
        hipLaunchKernelGGL(k_kd_dim, dim3((segs + 255) / 256), dim3(256), 0, s, bounds, segs, dim);
        hipLaunchKernelGGL(k_kd_keys, dim3(blocks), dim3(256), 0, s, pts, perm[cur], total, seglen, dim, keys[0]);
        size_t tb = temp_bytes;
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, tb, keys[0], keys[1], perm[cur], perm[cur ^ 1], (int)total, 0, (int)(32 + level), s);
        if (e != hipSuccess) return e;
        cur ^= 1;
        if (split_dim) hipLaunchKernelGGL(k_kd_split, dim3((segs + 255) / 256), dim3(256), 0, s, keys[1], dim, segs, seglen, split_dim, split_val);
    }
    hipLaunchKernelGGL(k_kd_gather, dim3((unsigned)(((size_t)total * 4 + 255) / 256)), dim3(256), 0, s, pts, perm[cur], total, index_base, out_points,
                       out_index);
    hipLaunchKernelGGL(k_kd_leaf_boxes, dim3((n_leaves * 16 + 255) / 256), dim3(256), 0, s, out_points, out_index, n_leaves, leaf_size, out_lo, out_hi);
    return hipGetLastError();
}

hipError_t kd_home_leaf(const float *qry, uint32_t n_qry, uint32_t n_leaves, const uint32_t *split_dim, const float *split_val, uint32_t *home,
                        hipStream_t s) {
    uint32_t depth = 0;
    while ((1u << depth) < n_leaves) ++depth;
    hipLaunchKernelGGL(k_kd_home, dim3((n_qry + 255) / 256), dim3(256), 0, s, qry, n_qry, depth, split_dim, split_val, home);
    return hipGetLastError();
}

}  // namespace rtdp
