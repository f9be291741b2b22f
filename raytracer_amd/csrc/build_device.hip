// Builder of the 4-wide hierarchy on the GPU (no counterpart in the reference, whose sources end at the datapath).
// A linear construction: triangles are ordered along a Morton curve through their centroids and every node
// quarters its range of that order by count, so the shape of the tree is known in advance:
//   level l holds 4^l nodes, node (l, j) covers sorted triangles [j n / 4^l, (j + 1) n / 4^l) and has the index
//   (4^l - 1) / 3 + j (children after parents, as the traversal's validation demands); below the last inner
//   level D - 1 the quarters are leaves, D chosen so that a leaf holds at most leaf_size triangles.
//   k_bvh_bounds   scene box of the centroids (ordered-integer atomics on the float images)
//   k_bvh_codes    30-bit Morton code per triangle (NaN / Inf centroids sort first)
//   (hipcub radix sort of (code, triangle) pairs)
//   k_bvh_gather   triangles into leaf order, 12 words each, and the order array
//   k_bvh_leaves   the nodes of level D - 1: exact boxes of their four leaves
//   k_bvh_level    levels D - 2 .. 0: a child's box is the union of that child's four boxes
// An empty range gives an absent child: reference 0xFFFFFFFF and the all-+Inf box of the reference's default
// box (src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:106-113).
#include <hipcub/hipcub.hpp>

#include "kernels.h"

namespace rtdp {

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu, kLeaf = 0x80000000u;

__device__ __forceinline__ uint32_t ordered(float f) {  // monotone map of the finite floats onto unsigned integers
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

__device__ __forceinline__ bool centroid(const float *__restrict__ t, float (&c)[3]) {
    bool finite = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = t[a], y = t[3 + a], z = t[6 + a];
        const float lo = fminf(fminf(x, y), z), hi = fmaxf(fmaxf(x, y), z);
        c[a] = 0.5f * lo + 0.5f * hi;
        finite = finite && (fabsf(c[a]) <= 3.0e38f) && (x == x) && (y == y) && (z == z);
    }
    return finite;
}

__global__ __launch_bounds__(256) void k_bvh_bounds(const float *__restrict__ tris, uint32_t n, uint32_t *__restrict__ bounds) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    float c[3];
    if (i < n && centroid(tris + (size_t)i * 9, c)) {
#pragma unroll
        for (int a = 0; a < 3; ++a) lo[a] = hi[a] = ordered(c[a]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t l = __shfl_down(lo[a], off), h = __shfl_down(hi[a], off);
            lo[a] = l < lo[a] ? l : lo[a];
            hi[a] = h > hi[a] ? h : hi[a];
        }
    }
    if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (lo[a] != 0xFFFFFFFFu) atomicMin(&bounds[a], lo[a]);
            if (hi[a] != 0u) atomicMax(&bounds[3 + a], hi[a]);
        }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {  // bit i of the low 10 bits -> bit 3 i
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void k_bvh_codes(const float *__restrict__ tris, uint32_t n, const uint32_t *__restrict__ bounds,
                                                  uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float c[3];
    uint32_t code = 0u;
    if (centroid(tris + (size_t)i * 9, c)) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = unordered(bounds[a]), hi = unordered(bounds[3 + a]);
            const float ext = hi - lo;
            float u = (ext > 0.0f && ext <= 3.0e38f) ? (c[a] - lo) / ext : 0.0f;
            u = (u >= 0.0f) ? (u <= 1.0f ? u : 1.0f) : 0.0f;
            uint32_t q = (uint32_t)(u * 1023.0f);
            q = q > 1023u ? 1023u : q;
            code |= spread10(q) << a;
        }
    }
    keys[i] = code;
    vals[i] = i;
}

__global__ __launch_bounds__(256) void k_bvh_gather(const float *__restrict__ tris, const uint32_t *__restrict__ perm, uint32_t n,
                                                   float *__restrict__ out_tris, uint32_t *__restrict__ out_order) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = perm[i];
    const float *t = tris + (size_t)src * 9;
    float *o = out_tris + (size_t)i * 12;
#pragma unroll
    for (int w = 0; w < 9; ++w) o[w] = t[w];
    o[9] = 0.0f; o[10] = 0.0f; o[11] = 0.0f;
    out_order[i] = src;
}

__device__ __forceinline__ uint32_t range_cut(uint32_t n, uint32_t j, uint32_t shift) {  // floor(j n / 4^level), 4^level = 1 << shift
    return (uint32_t)(((unsigned long long)j * n) >> shift);
}

// nodes of the last inner level (level D - 1, 4^(D-1) of them); their children are the leaves of level D
__global__ __launch_bounds__(256) void k_bvh_leaves(const float *__restrict__ tris12, uint32_t n, uint32_t level, uint32_t first_node,
                                                   float *__restrict__ nodes) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= (1u << (2u * level))) return;
    float *node = nodes + (size_t)(first_node + j) * 32;
    const uint32_t shift = 2u * (level + 1u);
    const float inf = __uint_as_float(0x7F800000u);
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
        const uint32_t a = range_cut(n, 4u * j + c, shift), b = range_cut(n, 4u * j + c + 1u, shift);
        float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        for (uint32_t t = a; t < b; ++t) {
            const float *v = tris12 + (size_t)t * 12;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float x = v[k];
                lo[k % 3] = (x < lo[k % 3]) ? x : lo[k % 3];
                hi[k % 3] = (x > hi[k % 3]) ? x : hi[k % 3];
            }
        }
        const bool some = b > a;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            node[6 * c + k] = some ? lo[k] : inf;
            node[6 * c + 3 + k] = some ? hi[k] : inf;
        }
        reinterpret_cast<uint32_t *>(node)[24 + c] = some ? (kLeaf | (a << 3) | (b - a - 1u)) : kNone;
    }
#pragma unroll
    for (int k = 28; k < 32; ++k) node[k] = 0.0f;
}

// nodes of an upper level: child (level + 1, 4 j + c) has its own four boxes, the child's box is their union
__global__ __launch_bounds__(256) void k_bvh_level(uint32_t n, uint32_t level, uint32_t first_node, uint32_t first_child, float *__restrict__ nodes) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= (1u << (2u * level))) return;
    float *node = nodes + (size_t)(first_node + j) * 32;
    const uint32_t shift = 2u * (level + 1u);
    const float inf = __uint_as_float(0x7F800000u);
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
        const uint32_t child = 4u * j + c;
        const bool some = range_cut(n, child + 1u, shift) > range_cut(n, child, shift);
        const float *cn = nodes + (size_t)(first_child + child) * 32;
        float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        if (some) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (reinterpret_cast<const uint32_t *>(cn)[24 + g] == kNone) continue;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float l = cn[6 * g + k], h = cn[6 * g + 3 + k];
                    lo[k] = (l < lo[k]) ? l : lo[k];
                    hi[k] = (h > hi[k]) ? h : hi[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            node[6 * c + k] = some ? lo[k] : inf;
            node[6 * c + 3 + k] = some ? hi[k] : inf;
        }
        reinterpret_cast<uint32_t *>(node)[24 + c] = some ? (first_child + child) : kNone;
    }
#pragma unroll
    for (int k = 28; k < 32; ++k) node[k] = 0.0f;
}

}  // namespace

// Depth D >= 1 with ceil(n / 4^D) <= leaf_size; the hierarchy has (4^D - 1) / 3 nodes.
uint32_t bvh_device_depth(uint32_t n_tris, uint32_t leaf_size) {
    uint32_t d = 1;
    while ((((unsigned long long)n_tris + (1ull << (2 * d)) - 1) >> (2 * d)) > leaf_size) ++d;
    return d;
}

size_t bvh_device_workspace(uint32_t n_tris) {
    size_t temp = 0;
    uint32_t *nul = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, nul, nul, nul, nul, (int)n_tris, 0, 30);
    return ((size_t)n_tris * 16 + 256 + temp + 255) & ~size_t(255);
}

hipError_t bvh_build_device(const float *tris, uint32_t n, uint32_t leaf_size, char *workspace, size_t workspace_bytes, float *out_nodes,
                            float *out_tris, uint32_t *out_order, hipStream_t s) {
    uint32_t *bounds = reinterpret_cast<uint32_t *>(workspace);
    uint32_t *keys = reinterpret_cast<uint32_t *>(workspace + 256);
    uint32_t *vals = keys + n, *keys2 = vals + n, *vals2 = keys2 + n;
    char *temp = reinterpret_cast<char *>(vals2 + n);
    size_t temp_bytes = workspace_bytes - (size_t)(temp - workspace);
    const unsigned blocks = (n + 255) / 256;
    static const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    hipError_t e = hipMemcpyAsync(bounds, init, sizeof init, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bvh_bounds, dim3(blocks), dim3(256), 0, s, tris, n, bounds);
    hipLaunchKernelGGL(k_bvh_codes, dim3(blocks), dim3(256), 0, s, tris, n, bounds, keys, vals);
    e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys2, vals, vals2, (int)n, 0, 30, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bvh_gather, dim3(blocks), dim3(256), 0, s, tris, vals2, n, out_tris, out_order);
    const uint32_t depth = bvh_device_depth(n, leaf_size);
    auto first_of = [](uint32_t level) { return (uint32_t)(((1ull << (2 * level)) - 1) / 3); };
    {
        const uint32_t level = depth - 1, count = 1u << (2 * level);
        hipLaunchKernelGGL(k_bvh_leaves, dim3((count + 255) / 256), dim3(256), 0, s, out_tris, n, level, first_of(level), out_nodes);
    }
    for (uint32_t level = depth - 1; level-- > 0;) {
        const uint32_t count = 1u << (2 * level);
        hipLaunchKernelGGL(k_bvh_level, dim3((count + 255) / 256), dim3(256), 0, s, n, level, first_of(level), first_of(level + 1), out_nodes);
    }
    return hipGetLastError();
}

}  // namespace rtdp
