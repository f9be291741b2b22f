// Traversal driver: the loop a ray tracer runs around the datapath's Quadbox and triangle
// ops (no counterpart in the reference's sources, which stop at the datapath; the op
// semantics it relies on are those of UnifiedDatapath.scala: children sorted by entry
// distance, hits first, :1117-1132, and the triangle test's hit / t_num / t_denom, :1087-1116).
//
// Breadth-first ("wavefront") over a 4-wide bounding volume hierarchy.  A level is
//   k_trace_box   one lane per (ray, node) pair: gather the ray and the node, run the
//                 Quadbox op, and turn the sorted hit mask into work for the next step.
//                 The hit children of all lanes of a wave are compacted with ballots (one
//                 ballot per sorted slot j gives "lanes with more than j hits"; the popcount
//                 below the lane is its offset) and appended with ONE atomic per wave and
//                 queue: inner children to the next level's queue, leaves to the leaf queue.
//   k_trace_tri   one lane per (ray, leaf) pair: the triangle op on each triangle of the
//                 leaf; a hit offers (t, triangle) to the ray's best[] with a 64-bit atomic
//                 min, t = t_num / t_denom being the division the RTL leaves to its consumer.
// The ray's extent entering a level is min(initial extent, best t after the previous
// level), so the set of (ray, node) pairs of every level, and with it the result, does not
// depend on the order in which waves append to the queues.
#include "ops.hpp"

#pragma clang fp contract(off)

namespace rtdp {

constexpr uint32_t kRefNone = 0xFFFFFFFFu;
constexpr uint32_t kRefLeaf = 0x80000000u;
constexpr int kRayWords = 16;   // origin 3, dir 3, inv 3, shear 3, k, extent, 2 unused
constexpr int kNodeWords = 32;  // 4 child boxes (min xyz, max xyz), 4 child refs, 4 unused
constexpr int kTriWords = 12;   // A, B, C, 3 unused
constexpr unsigned long long kBestNone = 0x7F800000FFFFFFFFull;  // (+Inf, no triangle)

// Every child reference must stay inside the arrays, and an inner child must have a larger
// index than its parent (which rules out cycles): checked on the device before any traversal,
// because a stray reference would be a wild gather.  bad[0] receives 1 + the first offender seen.
__global__ __launch_bounds__(256) void k_trace_validate(const float *__restrict__ nodes, uint32_t n_nodes, uint32_t n_tris,
                                                        uint32_t *__restrict__ bad) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_nodes) return;
    const u4v child = reinterpret_cast<const u4v *>(nodes + (size_t)i * kNodeWords)[6];
    const uint32_t c[4] = {child.x, child.y, child.z, child.w};
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (c[k] == kRefNone) continue;
        if (c[k] & kRefLeaf) {
            const uint32_t first = (c[k] & ~kRefLeaf) >> 3, cnt = (c[k] & 7u) + 1u;
            ok = ok && (first + cnt <= n_tris);
        } else {
            ok = ok && (c[k] > i) && (c[k] < n_nodes);
        }
    }
    if (!ok) atomicMax(bad, i + 1u);
}

__global__ __launch_bounds__(256) void k_trace_init(const float *ox, const float *oy, const float *oz, const float *dx, const float *dy,
                                                    const float *dz, uint32_t n, float extent, float *__restrict__ rays,
                                                    unsigned long long *__restrict__ best, uint2 *__restrict__ queue) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float d[3] = {dx[i], dy[i], dz[i]};
    float inv[3], shear[3];
    uint32_t k;
    ray_derive(d, inv, k, shear);
    f4v *r = reinterpret_cast<f4v *>(rays + (size_t)i * kRayWords);
    r[0] = f4v{ox[i], oy[i], oz[i], d[0]};
    r[1] = f4v{d[1], d[2], inv[0], inv[1]};
    r[2] = f4v{inv[2], shear[0], shear[1], shear[2]};
    r[3] = f4v{__uint_as_float(k), extent, 0.0f, 0.0f};
    best[i] = kBestNone;
    queue[i] = uint2{i, 0u};  // every ray starts at the root
}

// counters[0] = entries appended to `next`, counters[1] = to `leaf` (both count attempts, also
// beyond the capacity, so the host can see by how much a queue was too small).
__global__ __launch_bounds__(256) void k_trace_box(const uint2 *__restrict__ cur, uint32_t count, const float *__restrict__ rays,
                                                   const float *__restrict__ nodes, const unsigned long long *__restrict__ best,
                                                   uint2 *__restrict__ next, uint32_t next_cap, uint2 *__restrict__ leaf, uint32_t leaf_cap,
                                                   uint32_t *__restrict__ counters) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = i < count;
    uint32_t ray = 0, ref[4] = {kRefNone, kRefNone, kRefNone, kRefNone};
    uint32_t n_inner = 0, n_leaf = 0;
    if (live) {
        const uint2 item = cur[i];
        ray = item.x;
        const f4v *rr = reinterpret_cast<const f4v *>(rays + (size_t)ray * kRayWords);
        const f4v r0 = rr[0], r1 = rr[1], r2 = rr[2], r3 = rr[3];
        const f4v *nn = reinterpret_cast<const f4v *>(nodes + (size_t)item.y * kNodeWords);
        float r[kBoxFields];
        r[0] = r0.x; r[1] = r0.y; r[2] = r0.z; r[3] = r0.w; r[4] = r1.x; r[5] = r1.y; r[6] = r1.z; r[7] = r1.w; r[8] = r2.x;
        const float found = __uint_as_float((uint32_t)(best[ray] >> 32));  // +Inf while nothing was hit
        r[9] = (found < r3.y) ? found : r3.y;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const f4v v = nn[q];
            r[10 + 4 * q] = v.x; r[11 + 4 * q] = v.y; r[12 + 4 * q] = v.z; r[13 + 4 * q] = v.w;
        }
        const u4v child = reinterpret_cast<const u4v *>(nn)[6];
        float t[4];
        uint32_t flags;
        box_one(r, t, flags);
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // sorted slots: closest first, hits first
            const uint32_t slot = (flags >> (4 + 2 * k)) & 3u;
            const uint32_t c = (slot == 0u) ? child.x : (slot == 1u) ? child.y : (slot == 2u) ? child.z : child.w;
            const bool hit = ((flags >> k) & 1u) && c != kRefNone;
            ref[k] = hit ? c : kRefNone;
            n_inner += (hit && !(c & kRefLeaf)) ? 1u : 0u;
            n_leaf += (hit && (c & kRefLeaf)) ? 1u : 0u;
        }
    }
    // wave-level compaction: offset of this lane = entries of lower lanes
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t off_inner = 0, tot_inner = 0, off_leaf = 0, tot_leaf = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const unsigned long long mi = __ballot(n_inner > j), ml = __ballot(n_leaf > j);
        off_inner += (uint32_t)__popcll(mi & below); tot_inner += (uint32_t)__popcll(mi);
        off_leaf += (uint32_t)__popcll(ml & below);  tot_leaf += (uint32_t)__popcll(ml);
    }
    uint32_t base_inner = 0, base_leaf = 0;
    if (lane == 0) {
        if (tot_inner) base_inner = atomicAdd(&counters[0], tot_inner);
        if (tot_leaf) base_leaf = atomicAdd(&counters[1], tot_leaf);
    }
    base_inner = __builtin_amdgcn_readfirstlane(base_inner) + off_inner;
    base_leaf = __builtin_amdgcn_readfirstlane(base_leaf) + off_leaf;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (ref[k] == kRefNone) continue;
        if (ref[k] & kRefLeaf) {
            if (base_leaf < leaf_cap) leaf[base_leaf] = uint2{ray, ref[k]};
            ++base_leaf;
        } else {
            if (base_inner < next_cap) next[base_inner] = uint2{ray, ref[k]};
            ++base_inner;
        }
    }
}

__global__ __launch_bounds__(256) void k_trace_tri(const uint2 *__restrict__ leaf, uint32_t count, const float *__restrict__ rays,
                                                   const float *__restrict__ tris, unsigned long long *__restrict__ best,
                                                   uint32_t *__restrict__ counters) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint2 item = (i < count) ? leaf[i] : uint2{0u, 0u};
    {  // triangle ops evaluated by this wave (statistics only): bit-sliced ballot sum, one atomic per wave
        const uint32_t c = (i < count) ? (item.y & 7u) + 1u : 0u;
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b) sum += (uint32_t)__popcll(__ballot((c >> b) & 1u)) << b;
        if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&counters[2], sum);
    }
    if (i >= count) return;
    const f4v *rr = reinterpret_cast<const f4v *>(rays + (size_t)item.x * kRayWords);
    const f4v r0 = rr[0], r2 = rr[2], r3 = rr[3];
    const uint32_t first = (item.y & ~kRefLeaf) >> 3, cnt = (item.y & 7u) + 1u;
    float r[kTriFields];
    r[0] = r0.x; r[1] = r0.y; r[2] = r0.z; r[3] = r2.y; r[4] = r2.z; r[5] = r2.w;
    const uint32_t kw = __float_as_uint(r3.x);
    unsigned long long mine = kBestNone;
    for (uint32_t j = 0; j < cnt; ++j) {
        const f4v *tt = reinterpret_cast<const f4v *>(tris + (size_t)(first + j) * kTriWords);
        const f4v a = tt[0], b = tt[1], c = tt[2];
        r[6] = a.x; r[7] = a.y; r[8] = a.z; r[9] = a.w; r[10] = b.x; r[11] = b.y; r[12] = b.z; r[13] = b.w; r[14] = c.x;
        float num, den;
        uint32_t hit;
        tri_one(r, kw, num, den, hit);
        if (hit) {
            float t = num / den;  // IEEE division; den != 0 on a hit, num >= 0 or NaN
            if (t == t) {         // a NaN t (NaN operands) is no hit
                if (t == 0.0f) t = 0.0f;  // -0 -> +0: non-negative floats order like their bit patterns
                const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (first + j);
                mine = (key < mine) ? key : mine;
            }
        }
    }
    if (mine != kBestNone) atomicMin(best + item.x, mine);
}

// ---- depth-first variant: one lane walks one ray with a private stack ------------------
// The order the Quadbox op delivers (closest child first, hits first) is what a stack
// traversal wants: the hit children are pushed farthest first, so the nearest is popped
// next, and once a triangle is found every entry whose entry distance is not below the best
// t is dropped when popped.  Per ray, in this order:
//   pop (ref, tmin); limit = min(extent, best t); skip unless tmin < limit;
//   leaf: triangle op on each triangle in turn, a hit (t = t_num / t_denom, IEEE division, -0
//         read as +0, NaN ignored) replaces best when (t, triangle) is smaller;
//   node: Quadbox op with extent = limit; push the hit children in reverse sorted order.
// A ray's walk is sequential, so the result is deterministic by construction.
constexpr int kStackDepth = 64;
constexpr int kStackLds = 8;  // the top-of-stack entries that live in LDS ([entry][thread]: conflict-free); deeper ones in scratch

constexpr int kRefillLanes = 32;  // a wave fetches new rays once this many of its lanes have finished theirs

// Persistent form: the grid holds as many waves as the chip keeps resident; a lane that finished its ray takes
// another from a global counter (whole-wave refills, so the divergent set-up is paid rarely).  Which lane walks
// which ray is scheduling-dependent; what a ray's walk computes is not.
__global__ __launch_bounds__(256) void k_trace_dfs(const float *ox, const float *oy, const float *oz, const float *dx, const float *dy,
                                                   const float *dz, uint32_t n, float extent, const float *__restrict__ nodes,
                                                   const float *__restrict__ tris, float *__restrict__ out_t, uint32_t *__restrict__ out_tri,
                                                   unsigned long long *__restrict__ totals, uint32_t *__restrict__ overflow,
                                                   unsigned long long *__restrict__ next) {
    constexpr uint32_t kNoRay = 0xFFFFFFFFu;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lanes_total = (unsigned long long)gridDim.x * 256u;
    float r[kBoxFields], shear[3];
    uint32_t kw = 0;

    __shared__ uint2 s_stack[kStackLds][256];
    uint2 deep[kStackDepth - kStackLds];  // (reference, bits of the entry distance)
    int sp = 0;
    auto push = [&](uint2 e) {
        if (sp < kStackLds) s_stack[sp][threadIdx.x] = e;
        else deep[sp - kStackLds] = e;
        ++sp;
    };
    auto pop = [&]() -> uint2 {
        --sp;
        return (sp < kStackLds) ? s_stack[sp][threadIdx.x] : deep[sp - kStackLds];
    };
    uint32_t ray = kNoRay;
    unsigned long long best = kBestNone;
    auto take = [&](uint32_t id) {
        r[0] = ox[id]; r[1] = oy[id]; r[2] = oz[id];
        const float d[3] = {dx[id], dy[id], dz[id]};
        float inv[3];
        ray_derive(d, inv, kw, shear);
#pragma unroll
        for (int c = 0; c < 3; ++c) { r[3 + c] = d[c]; r[6 + c] = inv[c]; }
        ray = id;
        best = kBestNone;
        sp = 0;
        push(uint2{0u, 0u});  // the root, entry distance +0
    };
    {
        const unsigned long long first = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
        if (first < n) take((uint32_t)first);
    }
    bool exhausted = lanes_total >= n;  // wave-uniform: no ray left behind the counter
    uint32_t n_box = 0, n_tri = 0;
    bool spilled = false;

    for (;;) {
        if (ray != kNoRay && sp == 0) {  // this lane's ray is done
            out_t[ray] = __uint_as_float((uint32_t)(best >> 32));
            out_tri[ray] = (uint32_t)best;
            ray = kNoRay;
        }
        const unsigned long long idle = __ballot(ray == kNoRay);
        if (idle) {
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (!exhausted && n_idle >= (uint32_t)kRefillLanes) {
                const int leader = __ffsll((long long)idle) - 1;
                unsigned long long base = 0;
                if ((int)lane == leader) base = atomicAdd(next, (unsigned long long)n_idle);
                base = __shfl(base, leader) + lanes_total;
                if (ray == kNoRay) {
                    const unsigned long long id = base + (unsigned long long)__popcll(idle & ((1ull << lane) - 1ull));
                    if (id < n) take((uint32_t)id);
                }
                exhausted = base + n_idle >= n;
            }
            if (!__any(ray != kNoRay)) break;  // every lane idle and nothing left to fetch
        }
        uint32_t ref = kRefNone;
        float limit = 0.0f;
        if (ray != kNoRay) {
            const uint2 e = pop();
            const float found = __uint_as_float((uint32_t)(best >> 32));
            limit = (found < extent) ? found : extent;
            if (__uint_as_float(e.y) < limit) ref = e.x;
        }
        if (ref != kRefNone && !(ref & kRefLeaf)) {
            const f4v *nn = reinterpret_cast<const f4v *>(nodes + (size_t)ref * kNodeWords);
            r[9] = limit;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const f4v v = nn[q];
                r[10 + 4 * q] = v.x; r[11 + 4 * q] = v.y; r[12 + 4 * q] = v.z; r[13 + 4 * q] = v.w;
            }
            const u4v child = reinterpret_cast<const u4v *>(nn)[6];
            float t[4];
            uint32_t flags;
            box_one(r, t, flags);
            ++n_box;
#pragma unroll
            for (int k = 3; k >= 0; --k) {  // farthest first: the nearest child ends up on top
                const uint32_t slot = (flags >> (4 + 2 * k)) & 3u;
                const uint32_t c = (slot == 0u) ? child.x : (slot == 1u) ? child.y : (slot == 2u) ? child.z : child.w;
                if (((flags >> k) & 1u) && c != kRefNone) {
                    if (sp < kStackDepth) push(uint2{c, __float_as_uint(t[k])});
                    else spilled = true;
                }
            }
        } else if (ref != kRefNone) {
            const uint32_t first = (ref & ~kRefLeaf) >> 3, cnt = (ref & 7u) + 1u;
            float q[kTriFields];
            q[0] = r[0]; q[1] = r[1]; q[2] = r[2]; q[3] = shear[0]; q[4] = shear[1]; q[5] = shear[2];
            for (uint32_t j = 0; j < cnt; ++j) {
                const f4v *tt = reinterpret_cast<const f4v *>(tris + (size_t)(first + j) * kTriWords);
                const f4v a = tt[0], b = tt[1], c = tt[2];
                q[6] = a.x; q[7] = a.y; q[8] = a.z; q[9] = a.w; q[10] = b.x; q[11] = b.y; q[12] = b.z; q[13] = b.w; q[14] = c.x;
                float num, den;
                uint32_t hit;
                tri_one(q, kw, num, den, hit);
                if (hit) {
                    float t = num / den;
                    if (t == t) {
                        if (t == 0.0f) t = 0.0f;
                        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (first + j);
                        best = (key < best) ? key : best;
                    }
                }
            }
            n_tri += cnt;
        }
    }
    if (spilled) atomicOr(overflow, 1u);
    // statistics: one atomic pair per wave
    for (int off = 32; off > 0; off >>= 1) {
        n_box += __shfl_down(n_box, off);
        n_tri += __shfl_down(n_tri, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_box) atomicAdd(&totals[0], (unsigned long long)n_box);
        if (n_tri) atomicAdd(&totals[1], (unsigned long long)n_tri);
    }
}

__global__ __launch_bounds__(256) void k_trace_finish(const unsigned long long *__restrict__ best, uint32_t n, float *__restrict__ out_t,
                                                      uint32_t *__restrict__ out_tri) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const unsigned long long b = best[i];
    out_t[i] = __uint_as_float((uint32_t)(b >> 32));
    out_tri[i] = (uint32_t)b;
}

hipError_t trace_validate(const float *nodes, uint32_t n_nodes, uint32_t n_tris, uint32_t *bad, hipStream_t s) {
    hipLaunchKernelGGL(k_trace_validate, dim3((n_nodes + 255) / 256), dim3(256), 0, s, nodes, n_nodes, n_tris, bad);
    return hipGetLastError();
}

hipError_t trace_init(const float *const origin[3], const float *const dir[3], uint32_t n, float extent, float *rays,
                      unsigned long long *best, void *queue, hipStream_t s) {
    hipLaunchKernelGGL(k_trace_init, dim3((n + 255) / 256), dim3(256), 0, s, origin[0], origin[1], origin[2], dir[0], dir[1], dir[2], n, extent,
                       rays, best, static_cast<uint2 *>(queue));
    return hipGetLastError();
}

hipError_t trace_box(const void *cur, uint32_t count, const float *rays, const float *nodes, const unsigned long long *best, void *next,
                     uint32_t next_cap, void *leaf, uint32_t leaf_cap, uint32_t *counters, hipStream_t s) {
    hipLaunchKernelGGL(k_trace_box, dim3((count + 255) / 256), dim3(256), 0, s, static_cast<const uint2 *>(cur), count, rays, nodes, best,
                       static_cast<uint2 *>(next), next_cap, static_cast<uint2 *>(leaf), leaf_cap, counters);
    return hipGetLastError();
}

hipError_t trace_tri(const void *leaf, uint32_t count, const float *rays, const float *tris, unsigned long long *best, uint32_t *counters,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_trace_tri, dim3((count + 255) / 256), dim3(256), 0, s, static_cast<const uint2 *>(leaf), count, rays, tris, best,
                       counters);
    return hipGetLastError();
}

hipError_t trace_dfs(const float *const origin[3], const float *const dir[3], uint32_t n, float extent, const float *nodes, const float *tris,
                     float *out_t, uint32_t *out_tri, unsigned long long *totals, uint32_t *overflow, unsigned long long *next, hipStream_t s) {
    const uint32_t need = (n + 255) / 256, resident = 256u * 8u;  // 8 workgroups of 4 waves per CU stay resident
    hipLaunchKernelGGL(k_trace_dfs, dim3(need < resident ? need : resident), dim3(256), 0, s, origin[0], origin[1], origin[2], dir[0], dir[1],
                       dir[2], n, extent, nodes, tris, out_t, out_tri, totals, overflow, next);
    return hipGetLastError();
}

hipError_t trace_finish(const unsigned long long *best, uint32_t n, float *out_t, uint32_t *out_tri, hipStream_t s) {
    hipLaunchKernelGGL(k_trace_finish, dim3((n + 255) / 256), dim3(256), 0, s, best, n, out_t, out_tri);
    return hipGetLastError();
}

}  // namespace rtdp
