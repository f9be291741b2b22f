// Builder of the 4-wide hierarchy on the GPU (no counterpart in the reference, whose sources end at the datapath).
// A linear (LBVH-style) construction: the triangles are ordered along a Morton curve through their centroids,
// and a node's range of that order is halved twice where the highest differing bit of the codes flips, so the
// cuts are the cell boundaries of an octree; parts of at most leaf_size triangles become leaves.
//   k_bvh_bounds   scene box of the centroids (ordered-integer atomics on the float images)
//   k_bvh_codes    30-bit Morton code per triangle (NaN / Inf centroids sort first)
//   (hipcub radix sort of (code, triangle) pairs)
//   k_bvh_gather   triangles into leaf order, 12 words each, and the order array
//   top-down, one level per trip: k_lbvh_split (cut points, inner children per node), exclusive scan,
//   k_lbvh_emit (child references, the next level's ranges); nodes of a level are contiguous and follow their
//   parents, as the traversal's validation demands
//   bottom-up, deepest level first: k_lbvh_boxes (exact boxes of leaves, unions for inner children)
// An absent child has the reference 0xFFFFFFFF and the all-+Inf box of the reference's default box
// (src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:106-113).
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
            // thousands of waves would queue up on six words: go to the atomic only when the value read improves the bound
            if (lo[a] < __hip_atomic_load(&bounds[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&bounds[a], lo[a]);
            if (hi[a] > __hip_atomic_load(&bounds[3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&bounds[3 + a], hi[a]);
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

// First index of the upper part of sorted keys [x, y), y - x >= 2: where the highest bit in which the first and
// the last key differ flips (Karras' split), the midpoint when all keys are equal.  Both parts are non-empty.
__device__ __forceinline__ uint32_t split_point(const uint32_t *__restrict__ keys, uint32_t x, uint32_t y) {
    const uint32_t first = keys[x], last = keys[y - 1];
    if (first == last) return x + (y - x) / 2;
    const int common = __clz(first ^ last);
    uint32_t lo = x, hi = y - 1;  // keys[lo] shares more than `common` bits with first, keys[hi] does not
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (__clz(first ^ keys[mid]) > common) lo = mid;
        else hi = mid;
    }
    return hi;
}

// One level of the top-down pass.  ranges[i] = [a, b) of node first_node + i.  The range is halved twice at Morton
// split points; a part of at most leaf_size triangles becomes a leaf, a larger one an inner child that the next
// level will process.  Writes the three inner cut points and the number of inner children.
__global__ __launch_bounds__(256) void k_lbvh_split(const uint32_t *__restrict__ keys, const uint2 *__restrict__ ranges, uint32_t m,
                                                   uint32_t leaf_size, uint32_t *__restrict__ cuts, uint32_t *__restrict__ inner) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint32_t a = ranges[i].x, b = ranges[i].y;
    const uint32_t c2 = (b - a >= 2u) ? split_point(keys, a, b) : b;
    const uint32_t c1 = (c2 - a >= 2u) ? split_point(keys, a, c2) : c2;
    const uint32_t c3 = (b - c2 >= 2u) ? split_point(keys, c2, b) : b;
    const uint32_t c[5] = {a, c1, c2, c3, b};
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) cnt += (c[k + 1] - c[k] > leaf_size) ? 1u : 0u;
    cuts[3 * i] = c1; cuts[3 * i + 1] = c2; cuts[3 * i + 2] = c3;
    inner[i] = cnt;
}

// Child references of this level's nodes and the ranges of the next level (offsets = exclusive scan of `inner`).
__global__ __launch_bounds__(256) void k_lbvh_emit(const uint2 *__restrict__ ranges, uint32_t m, uint32_t leaf_size, const uint32_t *__restrict__ cuts,
                                                  const uint32_t *__restrict__ offsets, uint32_t first_node, uint32_t next_first,
                                                  float *__restrict__ nodes, uint2 *__restrict__ next_ranges) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint32_t c[5] = {ranges[i].x, cuts[3 * i], cuts[3 * i + 1], cuts[3 * i + 2], ranges[i].y};
    uint32_t *ref = reinterpret_cast<uint32_t *>(nodes + (size_t)(first_node + i) * 32) + 24;
    uint32_t slot = offsets[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = c[k], y = c[k + 1];
        if (y == x) ref[k] = kNone;
        else if (y - x <= leaf_size) ref[k] = kLeaf | (x << 3) | (y - x - 1u);
        else {
            ref[k] = next_first + slot;
            next_ranges[slot] = uint2{x, y};
            ++slot;
        }
    }
}

// Boxes of one level, deepest level first: a leaf child's box is that of its triangles, an inner child's box the
// union of the four boxes the child already holds.
__global__ __launch_bounds__(256) void k_lbvh_boxes(const float *__restrict__ tris12, uint32_t first_node, uint32_t m, float *__restrict__ nodes) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    float *node = nodes + (size_t)(first_node + i) * 32;
    const uint32_t *ref = reinterpret_cast<const uint32_t *>(node) + 24;
    const float inf = __uint_as_float(0x7F800000u);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t r = ref[c];
        float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        if (r != kNone && (r & kLeaf)) {
            const uint32_t a = (r & ~kLeaf) >> 3, cnt = (r & 7u) + 1u;
            for (uint32_t t = a; t < a + cnt; ++t) {
                const float *v = tris12 + (size_t)t * 12;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float x = v[k];
                    lo[k % 3] = (x < lo[k % 3]) ? x : lo[k % 3];
                    hi[k % 3] = (x > hi[k % 3]) ? x : hi[k % 3];
                }
            }
        } else if (r != kNone) {
            const float *cn = nodes + (size_t)r * 32;
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
            node[6 * c + k] = (r != kNone) ? lo[k] : inf;
            node[6 * c + 3 + k] = (r != kNone) ? hi[k] : inf;
        }
    }
#pragma unroll
    for (int k = 28; k < 32; ++k) node[k] = 0.0f;
}

}  // namespace

// Workspace: bounds, two (key, value) arrays, two range lists, cuts, inner counts and their scan, sort / scan scratch.
size_t bvh_device_workspace(uint32_t n_tris) {
    size_t sort_temp = 0, scan_temp = 0;
    uint32_t *nul = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_temp, nul, nul, nul, nul, (int)n_tris, 0, 30);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_temp, nul, nul, (int)n_tris);
    const size_t temp = sort_temp > scan_temp ? sort_temp : scan_temp;
    return ((size_t)n_tris * (16 + 16 + 12 + 8) + 512 + 256 + temp + 255) & ~size_t(255);  // 256: the sort / scan scratch starts on a 256-byte boundary
}

// Returns the number of nodes through *n_nodes (a host variable; at most n).  Synchronises the stream once per level.
hipError_t bvh_build_device(const float *tris, uint32_t n, uint32_t leaf_size, char *workspace, size_t workspace_bytes, float *out_nodes,
                            uint32_t node_capacity, uint32_t *n_nodes, float *out_tris, uint32_t *out_order, hipStream_t s) {
    uint32_t *bounds = reinterpret_cast<uint32_t *>(workspace);
    uint32_t *keys = reinterpret_cast<uint32_t *>(workspace + 512);
    uint32_t *vals = keys + n, *keys2 = vals + n, *vals2 = keys2 + n;
    uint2 *ranges[2] = {reinterpret_cast<uint2 *>(vals2 + n), reinterpret_cast<uint2 *>(vals2 + n) + n};
    uint32_t *cuts = reinterpret_cast<uint32_t *>(ranges[1] + n);
    uint32_t *inner = cuts + 3 * (size_t)n, *offsets = inner + n;
    // the library scratch of the radix sort / scan holds 8- and 16-byte records: start it on a 256-byte boundary
    // whatever the parity of n (the workspace itself comes from hipMalloc and is 256-byte aligned)
    char *temp = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(offsets + n) + 255u) & ~uintptr_t(255));
    const size_t temp_bytes = workspace_bytes - (size_t)(temp - workspace);
    const unsigned blocks = (n + 255) / 256;
    static const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    static const uint32_t zero = 0u;
    hipError_t e = hipMemcpyAsync(bounds, init, sizeof init, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bvh_bounds, dim3(blocks), dim3(256), 0, s, tris, n, bounds);
    hipLaunchKernelGGL(k_bvh_codes, dim3(blocks), dim3(256), 0, s, tris, n, bounds, keys, vals);
    size_t tb = temp_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(temp, tb, keys, keys2, vals, vals2, (int)n, 0, 30, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bvh_gather, dim3(blocks), dim3(256), 0, s, tris, vals2, n, out_tris, out_order);

    // top-down: level 0 is the root over [0, n) (also when n <= leaf_size: a root of up to four leaves)
    struct Level { uint32_t first, count; };
    Level levels[64];
    int depth = 0;
    e = hipMemcpyAsync(&ranges[0][0].x, &zero, 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&ranges[0][0].y, &n, 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // `n` lives on this stack frame
    if (e != hipSuccess) return e;
    uint32_t first = 0, count = 1;
    int cur = 0;
    while (count) {
        if (depth == 64 || (unsigned long long)first + count > node_capacity) return hipErrorInvalidValue;
        levels[depth++] = Level{first, count};
        const unsigned lb = (count + 255) / 256;
        hipLaunchKernelGGL(k_lbvh_split, dim3(lb), dim3(256), 0, s, keys2, ranges[cur], count, leaf_size, cuts, inner);
        tb = temp_bytes;
        e = hipcub::DeviceScan::ExclusiveSum(temp, tb, inner, offsets, (int)count, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_lbvh_emit, dim3(lb), dim3(256), 0, s, ranges[cur], count, leaf_size, cuts, offsets, first, first + count, out_nodes,
                           ranges[cur ^ 1]);
        uint32_t last[2];
        e = hipMemcpyAsync(&last[0], inner + (count - 1), 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&last[1], offsets + (count - 1), 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        first += count;
        count = last[0] + last[1];
        cur ^= 1;
    }
    *n_nodes = first;
    for (int l = depth - 1; l >= 0; --l)
        hipLaunchKernelGGL(k_lbvh_boxes, dim3((levels[l].count + 255) / 256), dim3(256), 0, s, out_tris, levels[l].first, levels[l].count, out_nodes);
    return hipGetLastError();
}

}  // namespace rtdp
