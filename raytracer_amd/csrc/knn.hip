// Exact k-nearest-neighbour search on the datapath's distance ops.  In this file:
//   k_knn_euclid16<false>  exhaustive scan ranked by the Euclidean op
//   k_knn_euclid16<true>   exhaustive scan ranked by the angular op's dot / sqrt(norm)
//   k_knn_tree16           the Euclidean search through a k-d index, wave by wave with box pruning
//   k_knn_pairs, k_knn_leaf_scan, k_knn_select, ...   the same search leaf by leaf
// All of them return the same lists for the same points.
//
// The distance between a query and a point (both 16 f32 = one beat) is exactly what
// the RTL delivers for a single-beat vector pair with mask 0xFFFF and reset 1:
// d_i = a_i - b_i, s_i = d_i * d_i, the adder tree (i, i+8), (i, i+4), (i, i+2), (0, 1)
// and finally accumulator(+0) + tree (reference src/main/scala/raytracer_datapath/
// UnifiedDatapath.scala:221-245, :399-419, :632-649, :783-803, :915-935, :1007-1027,
// :1133-1154), one IEEE RNE operation each, no fusion, canonical NaN.  Candidates are
// ranked by (distance, point index) with the RTL's compare semantics (lt / eq of
// CompareRecFN: NaN compares false), so a NaN distance never enters a list and the
// result does not depend on how the points are split over workgroups or GPUs.
//
// One lane owns one query (its 16 coordinates and its sorted top-16 list live in
// registers); the point index is wave-uniform, so the point's coordinates come in
// through the scalar data path (s_load) and feed the VALU as SGPR operands: no LDS,
// no barrier.  blockIdx.y splits the points so that a small query set still fills
// the chip; partial lists are merged afterwards (merge_keys in raytracer_amd/knn.py,
// the same merge that combines the per-GPU lists after the RCCL all-gather).
#include "kernels.h"

#pragma clang fp contract(off)

namespace rtdp {

constexpr int kTopK = 16;

typedef float f2v __attribute__((ext_vector_type(2)));

// One Euclidean beat with mask 0xFFFF entering a +0 accumulator.  Elements travel in pairs
// (2k, 2k+1) so that every subtract, multiply and tree add is one packed instruction
// (v_pk_add_f32 / v_pk_mul_f32: two independent IEEE operations, nothing fused).
__device__ __forceinline__ float knn_pair(const f2v (&qv)[8], const f2v *__restrict__ p) {
    f2v s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f2v d = qv[k] - p[k];
        s[k] = d * d;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = s[k] + s[k + 4];  // element i with i + 8
#pragma unroll
    for (int k = 0; k < 2; ++k) s[k] = s[k] + s[k + 2];  // i with i + 4
    s[0] = s[0] + s[1];                                   // i with i + 2
    // The accumulator step 0 + t and the NaN canonicalisation are omitted on purpose: t is a sum of
    // squares, never -0, so 0 + t == t bit for bit; and a NaN never passes the compares of
    // knn_offer, so no NaN reaches a list whatever its payload.
    return s[0].x + s[0].y;
}

// The angular op on a 16-element pair = two 8-element beats through the accumulators (reference
// src/main/scala/raytracer_datapath/UnifiedDatapath.scala:246-268 masking, :420-453 products, the adder tree
// (i, i+4), (i, i+2), (0, 1) of :650-676, :804-832, :936-964, accumulators :1028-1064):
//   dot = (0 + tree(q_i * c_i, i < 8)) + tree(q_i * c_i, i >= 8),  norm likewise over c_i * c_i,
// followed by what the RTL leaves to its consumer (IOSpec.scala:21-23): score = dot / sqrt(norm), both
// correctly rounded.  Returned negated, and -0 as +0, so that "larger score first" is the ascending order
// of the list code; a zero or NaN norm gives NaN, which no compare lets in.
__device__ __forceinline__ float knn_dot(const f2v (&qv)[8], const f2v *__restrict__ p) {
    float dot = 0.0f;
#pragma unroll
    for (int beat = 0; beat < 2; ++beat) {
        f2v s[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = qv[beat * 4 + k] * p[beat * 4 + k];
#pragma unroll
        for (int k = 0; k < 2; ++k) s[k] = s[k] + s[k + 2];  // element i with i + 4
        s[0] = s[0] + s[1];                                   // i with i + 2
        dot = dot + (s[0].x + s[0].y);
    }
    return dot;
}

// The norm half of the same two beats.  It depends on the candidate alone, so the search computes it once per
// point (one lane per point, 64 points at a time) and hands it to the wave by lane reads.
__device__ __forceinline__ float knn_norm(const f2v (&c)[8]) {
    float norm = 0.0f;
#pragma unroll
    for (int beat = 0; beat < 2; ++beat) {
        f2v m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = c[beat * 4 + k] * c[beat * 4 + k];
#pragma unroll
        for (int k = 0; k < 2; ++k) m[k] = m[k] + m[k + 2];
        m[0] = m[0] + m[1];
        norm = norm + (m[0].x + m[0].y);
    }
    return norm;
}

__device__ __forceinline__ float knn_neg_score(float dot, float norm) {
    const float neg = -(dot / sqrtf(norm));
    return (neg == 0.0f) ? 0.0f : neg;
}

// One pass of insertion of (cd, ci) into the ascending list; a NaN candidate changes nothing.
__device__ __forceinline__ void knn_insert(float (&bd)[kTopK], uint32_t (&bi)[kTopK], float cd, uint32_t ci) {
#pragma unroll
    for (int t = 0; t < kTopK; ++t) {
        const bool lt = (cd < bd[t]) || (cd == bd[t] && ci < bi[t]);  // lt, or eq with a smaller index
        const float nd = lt ? cd : bd[t];
        const uint32_t ni = lt ? ci : bi[t];
        cd = lt ? bd[t] : cd;
        ci = lt ? bi[t] : ci;
        bd[t] = nd;
        bi[t] = ni;
    }
}

// Offers (dist, idx) to the list.  An insertion pass costs every lane of the wave ~120
// instructions, whichever lanes need it, so a lane parks its candidate in (pd, pi) and the
// wave runs one pass only when some lane comes up with a second one: a pass then serves
// many lanes at once.  The list is the set of the 16 smallest (dist, idx) keys offered, so
// the order in which candidates are inserted is irrelevant; a parked candidate only delays
// the tightening of its lane's filter bound, never the result.  pd is NaN when empty.
// thr is the lane's filter bound: never below the distance of the 16th-best key that the
// final list can hold, so dropping dist > thr loses nothing (dist == thr still passes: the
// index decides).
__device__ __forceinline__ void knn_offer(float (&bd)[kTopK], uint32_t (&bi)[kTopK], float &pd, uint32_t &pi, float &thr, float dist, uint32_t idx) {
    const bool pass = dist <= thr;  // one compare filters almost every candidate (false for NaN)
    if (__any(pass)) {
        if (__any(pass && (pd == pd))) {
            knn_insert(bd, bi, pd, pi);
            pd = __uint_as_float(0x7FC00000u);
            thr = (bd[kTopK - 1] < thr) ? bd[kTopK - 1] : thr;
        }
        pd = pass ? dist : pd;
        pi = pass ? idx : pi;
    }
}

// The angular search's offer.  The exact key -RN(dot / RN(sqrt(norm))) costs two correctly rounded
// operations; almost every pair is turned away first by the estimate est = -dot * rsq(norm) (v_rsq_f32, 1 ulp),
// whose relative distance from the exact key is below 2^-21.  A pair is examined unless
//     est - |est| 2^-19  >  gate,   gate = thr + |thr| 2^-19 + 1e-30
// (the margins widen both sides by 2^-19 relative, the absolute term covers products that underflow), written
// as a negated "greater than" so that a NaN estimate is examined too; rsq_norm arrives as NaN for a point whose
// norm is so small that the estimate's input may have been flushed.  The examined pairs go through knn_offer
// with their exact keys, so the lists hold exact keys only.  `gate` follows thr, which only knn_offer changes.
__device__ __forceinline__ float knn_gate(float thr) { return thr + fabsf(thr) * 1.9073486328125e-06f + 1.0e-30f; }

__device__ __forceinline__ void knn_offer_angular(float (&bd)[kTopK], uint32_t (&bi)[kTopK], float &pd, uint32_t &pi, float &thr, float &gate,
                                                  float dot, float norm, float rsq_norm, uint32_t idx) {
    const float est = -(dot * rsq_norm);
    const bool maybe = !(est - fabsf(est) * 1.9073486328125e-06f > gate);
    if (__any(maybe)) {
        knn_offer(bd, bi, pd, pi, thr, maybe ? knn_neg_score(dot, norm) : __uint_as_float(0x7FC00000u), idx);
        gate = knn_gate(thr);
    }
}

__device__ __forceinline__ float lane_read(float v, uint32_t lane) {  // lane is wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)lane));
}

constexpr int kKnnUnroll = 4;  // points per loop trip: their scalar loads are issued together

template <bool ANG>
__global__ __launch_bounds__(256) void k_knn_euclid16(const float *__restrict__ pts, uint32_t n_pts, const float *__restrict__ qry,
                                                     uint32_t n_qry, uint32_t idx_base, uint32_t pts_per_split,
                                                     float *__restrict__ out_d, uint32_t *__restrict__ out_i) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    const bool live = q < n_qry;
    const uint32_t qq = live ? q : (n_qry - 1);
    f2v qv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qv[k] = reinterpret_cast<const f2v *>(qry)[(size_t)qq * 8 + k];

    float bd[kTopK];
    uint32_t bi[kTopK];
#pragma unroll
    for (int j = 0; j < kTopK; ++j) { bd[j] = __uint_as_float(0x7F800000u); bi[j] = 0xFFFFFFFFu; }
    float pd = __uint_as_float(0x7FC00000u);  // parked candidate (none)
    uint32_t pi = 0;
    float thr = __uint_as_float(0x7F800000u);

    const uint32_t j0 = blockIdx.y * pts_per_split;
    const uint32_t j1 = (j0 + pts_per_split < n_pts) ? j0 + pts_per_split : n_pts;
    if (ANG) {
        const uint32_t lane = threadIdx.x & 63u;
        float gate = knn_gate(thr);
        for (uint32_t jb = j0; jb < j1; jb += 64u) {
            const uint32_t cnt = (j1 - jb < 64u) ? (j1 - jb) : 64u;
            // this lane's point of the block: its norm and the reciprocal square root estimate of the filter
            const f2v *mine = reinterpret_cast<const f2v *>(pts) + (size_t)(jb + (lane < cnt ? lane : cnt - 1u)) * 8;
            f2v c[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = mine[k];
            const float my_norm = knn_norm(c);
            const float my_rsq = (my_norm < 1.0e-30f) ? __uint_as_float(0x7FC00000u) : __builtin_amdgcn_rsqf(my_norm);
            uint32_t u = 0;
            for (; u + kKnnUnroll <= cnt; u += kKnnUnroll) {
                const f2v *p = reinterpret_cast<const f2v *>(pts) + (size_t)(jb + u) * 8;  // wave-uniform address
                float dot[kKnnUnroll];
#pragma unroll
                for (int v = 0; v < kKnnUnroll; ++v) dot[v] = knn_dot(qv, p + v * 8);
#pragma unroll
                for (int v = 0; v < kKnnUnroll; ++v)
                    knn_offer_angular(bd, bi, pd, pi, thr, gate, dot[v], lane_read(my_norm, u + v), lane_read(my_rsq, u + v), idx_base + jb + u + v);
            }
            for (; u < cnt; ++u) {
                const f2v *p = reinterpret_cast<const f2v *>(pts) + (size_t)(jb + u) * 8;
                knn_offer_angular(bd, bi, pd, pi, thr, gate, knn_dot(qv, p), lane_read(my_norm, u), lane_read(my_rsq, u), idx_base + jb + u);
            }
        }
    } else {
        uint32_t j = j0;
        for (; j + kKnnUnroll <= j1; j += kKnnUnroll) {
            const f2v *p = reinterpret_cast<const f2v *>(pts) + (size_t)j * 8;  // wave-uniform address
            float dist[kKnnUnroll];
#pragma unroll
            for (int u = 0; u < kKnnUnroll; ++u) dist[u] = knn_pair(qv, p + u * 8);
#pragma unroll
            for (int u = 0; u < kKnnUnroll; ++u) knn_offer(bd, bi, pd, pi, thr, dist[u], idx_base + j + u);
        }
        for (; j < j1; ++j) {
            const f2v *p = reinterpret_cast<const f2v *>(pts) + (size_t)j * 8;
            knn_offer(bd, bi, pd, pi, thr, knn_pair(qv, p), idx_base + j);
        }
    }
    knn_insert(bd, bi, pd, pi);
    if (live) {
        const size_t o = ((size_t)blockIdx.y * n_qry + q) * kTopK;
#pragma unroll
        for (int t = 0; t < kTopK; ++t) { out_d[o + t] = bd[t]; out_i[o + t] = bi[t]; }
    }
}

// ---- the same search through a tree over the points ---------------------------------
// The points are stored leaf by leaf (a balanced k-d partition built by the caller: 2^D
// leaves of `leaf_size` slots, pad slots at +Inf with index 0xFFFFFFFF) with one
// 16-dimensional bounding box per leaf.  A wave (64 queries, sorted by home leaf so that
// they are neighbours in space) walks the leaves nearest-subtree-first (home ^ t) and scans
// a leaf only if some lane's lower bound on the distance to its box does not exceed that
// lane's filter bound.  Whatever is scanned goes through the very same distance and list
// code as the exhaustive kernel, so the result is bit-identical to it provided no pruned
// leaf holds a point that belongs in a list:
//   * in exact arithmetic  B = sum_d max(lo_d - q_d, q_d - hi_d, 0)^2  <=  |q - p|^2  for
//     every p in the box; the datapath's value of |q - p|^2 and the computed B each carry a
//     relative rounding error below 2^-19 (at most 16 roundings on non-negative terms), and
//     terms that underflow add at most 16 * 2^-126 absolutely; the test uses
//     B * (1 - 2^-16) - 2^-122, which is therefore below every computed distance in the leaf;
//   * a NaN in the bound arithmetic (Inf - Inf) reads as "no gap" (fmaxf drops it); boxes of
//     leaves holding NaN coordinates are unbounded in that dimension (caller's contract).
// blockIdx.y = s takes the leaves whose low log2(splits) index bits equal s: every split
// starts next to the home leaf.  The splits of one query share their progress through
// bound[q] (bits of a non-negative float, atomic min): it only ever holds the distance of
// some list's 16th entry, an upper bound of the final 16th distance, so using it as the
// filter bound changes how much is scanned, never what is found.
struct KnnTreeArgs {
    const float *pts;
    const uint32_t *index;
    const float *leaf_lo, *leaf_hi;
    uint32_t n_leaves, leaf_size, log2_splits;
    uint32_t walk;  // leaves visited per split, nearest subtree first (0 = all of them)
};

__global__ __launch_bounds__(256) void k_knn_tree16(KnnTreeArgs tree, const float *__restrict__ qry, const uint32_t *__restrict__ home,
                                                   uint32_t n_qry, uint32_t *__restrict__ bound, float *__restrict__ out_d,
                                                   uint32_t *__restrict__ out_i) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    const bool live = q < n_qry;
    const uint32_t qq = live ? q : (n_qry - 1);
    f2v qv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qv[k] = reinterpret_cast<const f2v *>(qry)[(size_t)qq * 8 + k];

    float bd[kTopK];
    uint32_t bi[kTopK];
#pragma unroll
    for (int j = 0; j < kTopK; ++j) { bd[j] = __uint_as_float(0x7F800000u); bi[j] = 0xFFFFFFFFu; }
    float pd = __uint_as_float(0x7FC00000u);
    uint32_t pi = 0;
    float thr = __uint_as_float(0x7F800000u);
    float published = thr;

    const uint32_t split = blockIdx.y;
    const uint32_t per_split = tree.n_leaves >> tree.log2_splits;
    // the wave follows its first query; the value is reduced into the leaf range, so no caller-supplied number can
    // steer the walk outside the index
    const uint32_t hh = (__builtin_amdgcn_readfirstlane(home[qq]) >> tree.log2_splits) & (per_split - 1u);
    const f2v *pts2 = reinterpret_cast<const f2v *>(tree.pts);

    const uint32_t n_walk = (tree.walk && tree.walk < per_split) ? tree.walk : per_split;
    for (uint32_t t = 0; t < n_walk; ++t) {
        const uint32_t leaf = ((hh ^ t) << tree.log2_splits) | split;  // wave-uniform
        const f2v *lo = reinterpret_cast<const f2v *>(tree.leaf_lo) + (size_t)leaf * 8;
        const f2v *hi = reinterpret_cast<const f2v *>(tree.leaf_hi) + (size_t)leaf * 8;
        const float shared = __uint_as_float(__hip_atomic_load(bound + qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        float lb = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f2v a = lo[k] - qv[k], b = qv[k] - hi[k];
            const float gx = fmaxf(fmaxf(a.x, b.x), 0.0f), gy = fmaxf(fmaxf(a.y, b.y), 0.0f);
            lb += gx * gx;
            lb += gy * gy;
        }
        thr = (shared < thr) ? shared : thr;
        const float lbd = lb * 0.9999847412109375f - 1.8807909613156600e-37f;  // (1 - 2^-16), 2^-122
        if (!__any(lbd <= thr)) continue;

        const f2v *p = pts2 + (size_t)leaf * tree.leaf_size * 8;
        const uint32_t *ix = tree.index + (size_t)leaf * tree.leaf_size;
        for (uint32_t j = 0; j < tree.leaf_size; j += kKnnUnroll) {
            float dist[kKnnUnroll];
#pragma unroll
            for (int u = 0; u < kKnnUnroll; ++u) dist[u] = knn_pair(qv, p + (size_t)(j + u) * 8);
#pragma unroll
            for (int u = 0; u < kKnnUnroll; ++u) knn_offer(bd, bi, pd, pi, thr, dist[u], ix[j + u]);
        }
        if (live && bd[kTopK - 1] < published) {
            published = bd[kTopK - 1];
            atomicMin(bound + q, __float_as_uint(published));
        }
    }
    knn_insert(bd, bi, pd, pi);
    if (live) {
        const size_t o = ((size_t)blockIdx.y * n_qry + q) * kTopK;
#pragma unroll
        for (int t = 0; t < kTopK; ++t) { out_d[o + t] = bd[t]; out_i[o + t] = bi[t]; }
    }
}

hipError_t launch_knn_tree16(const float *pts, const uint32_t *index, const float *leaf_lo, const float *leaf_hi, uint32_t n_leaves,
                             uint32_t leaf_size, const float *qry, const uint32_t *home, uint32_t n_qry, uint32_t log2_splits, uint32_t walk,
                             uint32_t *bound, float *out_d, uint32_t *out_i, hipStream_t stream) {
    if (n_qry == 0) return hipSuccess;
    hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(bound), 0x7F800000, n_qry, stream);  // +Inf: nothing known yet
    if (e != hipSuccess) return e;
    KnnTreeArgs tree{pts, index, leaf_lo, leaf_hi, n_leaves, leaf_size, log2_splits, walk};
    dim3 grid((n_qry + 255) / 256, 1u << log2_splits);
    hipLaunchKernelGGL(k_knn_tree16, grid, dim3(256), 0, stream, tree, qry, home, n_qry, bound, out_d, out_i);
    return hipGetLastError();
}

// ---- the angular search through the same tree ---------------------------------------------
// Largest score = dot / sqrt(norm) first.  A leaf is skipped only when NO point inside its box can reach a lane's list:
// with U an upper bound of every COMPUTED score in the box, the leaf is scanned unless -U > thr for every lane (thr = the
// lane's 16th best negated score so far, or the one its splits share).  What is scanned goes through the exhaustive
// kernel's code (norm once per point, the rsq filter, exact keys), so the lists are those of k_knn_euclid16<true>.
//
// U, per query q and box [lo, hi] (float arithmetic; its own rounding, at most 2^-20 relative on sums of 16 terms, is
// covered by the margins):
//   D  = sum_i max(q_i lo_i, q_i hi_i)   >= the exact dot product with any c in the box
//   A  = sum_i max(|q_i lo_i|, |q_i hi_i|) >= sum_i |q_i c_i|: the datapath's dot (16 rounded products, adder trees and
//        accumulator steps at most 6 additions deep, :420-453, :650-676 ... :1028-1064) differs from the exact one by less
//        than 2^-21 A, plus at most 2^-140 for products that lose bits in the subnormal range
//   m2 = sum_i min(c_i^2 : lo_i <= c_i <= hi_i),  M2 = sum_i max(lo_i^2, hi_i^2): the exact norm lies between them and the
//        datapath's norm (non-negative terms) within 2^-21 relative and 2^-140 absolute of it
//   the two correctly rounded operations of the consumer step (IOSpec.scala:21-23) add 2^-23 relative.
//   num = D + 2^-17 A + 1e-36;   num >= 0:  U = num / sqrt(m2 (1 - 2^-17) - 1e-36) * (1 + 2^-17) + 1e-36   (+Inf when the
//   root's argument is not positive: a norm that may underflow to 0 makes the score infinite);   num < 0:  U = num /
//   sqrt(M2 (1 + 2^-17) + 1e-36) * (1 - 2^-17)   (the absolute terms also cover squares and products of the bound's own
//   arithmetic that underflow, 16 x 2^-126 at most).
// Where float overflow could break these relations (A or M2 beyond 1e37), where the box is unbounded in a dimension the
// query uses (0 x Inf = NaN), or where a leaf holds NaN coordinates (box unbounded by the caller's contract), the bound
// comes out +Inf or NaN and the test, written as a negated compare, scans the leaf.  An empty box (lo > hi: a leaf of pad
// slots only) is skipped: its slots produce NaN scores, which no list takes.
//
// The splits of one query share thr through bound[q] as the order-preserving integer image of the float (negative:
// all bits flipped, otherwise sign bit set), combined with an atomic min.
__device__ __forceinline__ uint32_t ordered_bits(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

__global__ __launch_bounds__(256) void k_knn_tree_ang16(KnnTreeArgs tree, const float *__restrict__ qry, const uint32_t *__restrict__ home,
                                                       uint32_t n_qry, uint32_t *__restrict__ bound, float *__restrict__ out_d,
                                                       uint32_t *__restrict__ out_i, unsigned long long *__restrict__ scanned) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool live = q < n_qry;
    const uint32_t qq = live ? q : (n_qry - 1);
    f2v qv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qv[k] = reinterpret_cast<const f2v *>(qry)[(size_t)qq * 8 + k];

    float bd[kTopK];
    uint32_t bi[kTopK];
#pragma unroll
    for (int j = 0; j < kTopK; ++j) { bd[j] = __uint_as_float(0x7F800000u); bi[j] = 0xFFFFFFFFu; }
    float pd = __uint_as_float(0x7FC00000u);
    uint32_t pi = 0;
    float thr = __uint_as_float(0x7F800000u);
    float gate = knn_gate(thr);
    float published = thr;

    const uint32_t split = blockIdx.y;
    const uint32_t per_split = tree.n_leaves >> tree.log2_splits;
    const uint32_t hh = (__builtin_amdgcn_readfirstlane(home[qq]) >> tree.log2_splits) & (per_split - 1u);
    const f2v *pts2 = reinterpret_cast<const f2v *>(tree.pts);
    uint32_t n_scanned = 0;

    for (uint32_t t = 0; t < per_split; ++t) {
        const uint32_t leaf = ((hh ^ t) << tree.log2_splits) | split;  // wave-uniform
        const float *lo = tree.leaf_lo + (size_t)leaf * 16, *hi = tree.leaf_hi + (size_t)leaf * 16;
        const float shared = ordered_float(__hip_atomic_load(bound + qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (shared < thr) { thr = shared; gate = knn_gate(thr); }
        float D = 0.0f, A = 0.0f, m2 = 0.0f, M2 = 0.0f;
        bool empty = false;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float l = lo[i], h = hi[i], x = (i & 1) ? qv[i >> 1].y : qv[i >> 1].x;
            empty |= l > h;
            const float pl = x * l, ph = x * h;
            D += fmaxf(pl, ph);
            A += fmaxf(fabsf(pl), fabsf(ph));
            const float ll = l * l, hq = h * h;
            m2 += (l > 0.0f) ? ll : ((h < 0.0f) ? hq : 0.0f);
            M2 += fmaxf(ll, hq);
        }
        constexpr float kEps = 7.62939453125e-06f, kAbs = 1.0e-36f;  // 2^-17
        const float num = D + A * kEps + kAbs;
        float U;
        if (num >= 0.0f) {
            const float den2 = m2 * (1.0f - kEps) - kAbs;
            U = (den2 > 0.0f) ? num / sqrtf(den2) * (1.0f + kEps) + kAbs : __uint_as_float(0x7F800000u);
        } else {
            U = num / sqrtf(M2 * (1.0f + kEps) + kAbs) * (1.0f - kEps);
        }
        const bool safe = (A < 1.0e37f) && (M2 < 1.0e37f);  // false for NaN as well
        const bool skip = empty || (safe && (-U > thr));
        if (!__any(!skip)) continue;
        ++n_scanned;

        const uint32_t *ix = tree.index + (size_t)leaf * tree.leaf_size;
        for (uint32_t jb = 0; jb < tree.leaf_size; jb += 64u) {
            const uint32_t cnt = (tree.leaf_size - jb < 64u) ? (tree.leaf_size - jb) : 64u;  // a multiple of 4
            // this lane's point of the block: its norm and the reciprocal square root estimate of the filter
            const f2v *mine = pts2 + ((size_t)leaf * tree.leaf_size + jb + (lane < cnt ? lane : cnt - 1u)) * 8;
            f2v c[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = mine[k];
            const float my_norm = knn_norm(c);
            const float my_rsq = (my_norm < 1.0e-30f) ? __uint_as_float(0x7FC00000u) : __builtin_amdgcn_rsqf(my_norm);
            for (uint32_t u = 0; u < cnt; u += kKnnUnroll) {
                const f2v *p = pts2 + ((size_t)leaf * tree.leaf_size + jb + u) * 8;  // wave-uniform address
                float dot[kKnnUnroll];
#pragma unroll
                for (int v = 0; v < kKnnUnroll; ++v) dot[v] = knn_dot(qv, p + v * 8);
#pragma unroll
                for (int v = 0; v < kKnnUnroll; ++v)
                    knn_offer_angular(bd, bi, pd, pi, thr, gate, dot[v], lane_read(my_norm, u + v), lane_read(my_rsq, u + v), ix[jb + u + v]);
            }
        }
        if (live && bd[kTopK - 1] < published) {
            published = bd[kTopK - 1];
            atomicMin(bound + q, ordered_bits(published));
        }
    }
    knn_insert(bd, bi, pd, pi);
    if (live) {
        const size_t o = ((size_t)blockIdx.y * n_qry + q) * kTopK;
#pragma unroll
        for (int t = 0; t < kTopK; ++t) { out_d[o + t] = bd[t]; out_i[o + t] = bi[t]; }
    }
    if (scanned && lane == 0) atomicAdd(scanned, (unsigned long long)n_scanned);  // leaves scanned, summed over waves (statistics)
}

hipError_t launch_knn_tree_ang16(const float *pts, const uint32_t *index, const float *leaf_lo, const float *leaf_hi, uint32_t n_leaves,
                                 uint32_t leaf_size, const float *qry, const uint32_t *home, uint32_t n_qry, uint32_t log2_splits,
                                 uint32_t *bound, float *out_d, uint32_t *out_i, unsigned long long *scanned, hipStream_t stream) {
    if (n_qry == 0) return hipSuccess;
    hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(bound), 0xFF800000u, n_qry, stream);  // ordered image of +Inf: nothing known yet
    if (e != hipSuccess) return e;
    KnnTreeArgs tree{pts, index, leaf_lo, leaf_hi, n_leaves, leaf_size, log2_splits, 0u};
    dim3 grid((n_qry + 255) / 256, 1u << log2_splits);
    hipLaunchKernelGGL(k_knn_tree_ang16, grid, dim3(256), 0, stream, tree, qry, home, n_qry, bound, out_d, out_i, scanned);
    return hipGetLastError();
}

// ---- leaf-major search ------------------------------------------------------------------
// Where neighbouring queries do not share their candidate leaves (uniform data in 16
// dimensions: a query needs ~27 % of the leaves, 64 neighbouring queries together 99 %),
// pruning per wave of queries saves nothing.  The leaf-major form turns the loop around:
//   1. bound     k_knn_tree16 over the few leaves nearest to each query leaves tau_q, the
//                16th distance found so far: an upper bound of the final 16th distance.
//   2. lists     k_knn_pairs (count, then fill): for every leaf the queries whose deflated
//                box bound does not exceed their tau_q, compacted per wave with a ballot.
//   3. scan      k_knn_leaf_scan: one wave per 64 queries of one leaf's list; the leaf's
//                points arrive as scalar operands exactly as in the kernels above; every
//                (distance, index) with distance <= tau_q goes to that query's candidate
//                slots (one atomic counter per query).
//   4. select    k_knn_select: one wave per query keeps the 16 smallest candidate keys.
// Every point within tau_q of its query lies in a listed leaf (same box-bound argument as
// above) and passes the distance filter, and the final list is made of points within the
// final 16th distance <= tau_q: the result is the exhaustive result, bit for bit.  A query
// that collects more candidates than it has slots is flagged for the exhaustive kernel.
// Two passes over (query, leaf): the first counts per leaf, the second, with count[] restarted from 0 as the
// cursor, fills the lists.  `subset` (or NULL) names the queries of a later round.  `masks` (or NULL), one
// 64-bit ballot per wave and leaf, carries the first pass's decisions to the second, which then skips the box test.
__global__ __launch_bounds__(256) void k_knn_pairs(KnnTreeArgs tree, const float *__restrict__ qry, const uint32_t *__restrict__ subset,
                                                  uint32_t n_qry, const uint32_t *__restrict__ tau, uint32_t *__restrict__ count,
                                                  const uint32_t *__restrict__ offset, uint32_t *__restrict__ list,
                                                  unsigned long long *__restrict__ masks) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = i < n_qry;
    const uint32_t q = subset ? subset[live ? i : (n_qry - 1)] : (live ? i : (n_qry - 1));
    const uint32_t l0 = blockIdx.y * (tree.n_leaves / gridDim.y), l1 = l0 + tree.n_leaves / gridDim.y;
    // a wave that lies entirely behind the last query owns no row of the ballot array
    unsigned long long *wave_masks = (masks && (i & ~63u) < n_qry) ? masks + (size_t)(i >> 6) * tree.n_leaves : nullptr;
    if (masks && !wave_masks) return;
    if (list && masks) {  // second pass with remembered decisions
        for (uint32_t leaf = l0; leaf < l1; ++leaf) {
            const unsigned long long m = wave_masks[leaf];  // wave-uniform
            if (!m) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(count + leaf, (uint32_t)__popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if ((m >> lane) & 1ull) list[offset[leaf] + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = q;
        }
        return;
    }
    f2v qv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qv[k] = reinterpret_cast<const f2v *>(qry)[(size_t)q * 8 + k];
    const float thr = __uint_as_float(tau[q]);
    for (uint32_t leaf = l0; leaf < l1; ++leaf) {
        const f2v *lo = reinterpret_cast<const f2v *>(tree.leaf_lo) + (size_t)leaf * 8;
        const f2v *hi = reinterpret_cast<const f2v *>(tree.leaf_hi) + (size_t)leaf * 8;
        float lb = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f2v a = lo[k] - qv[k], b = qv[k] - hi[k];
            const float gx = fmaxf(fmaxf(a.x, b.x), 0.0f), gy = fmaxf(fmaxf(a.y, b.y), 0.0f);
            lb += gx * gx;
            lb += gy * gy;
        }
        const float lbd = lb * 0.9999847412109375f - 1.8807909613156600e-37f;  // (1 - 2^-16), 2^-122
        const bool need = live && (lbd <= thr);
        const unsigned long long m = __ballot(need);
        if (wave_masks && lane == 0) wave_masks[leaf] = m;
        if (!m) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(count + leaf, (uint32_t)__popcll(m));
        if (list) {
            base = __builtin_amdgcn_readfirstlane(base);
            if (need) list[offset[leaf] + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = q;
        }
    }
}

// group_leaf[g] = the leaf of work group g; groups of one leaf are consecutive, group_first[leaf] is its first.
__global__ __launch_bounds__(64) void k_knn_leaf_scan(KnnTreeArgs tree, const float *__restrict__ qry, const uint32_t *__restrict__ tau,
                                                     const uint32_t *__restrict__ group_leaf, const uint32_t *__restrict__ group_first,
                                                     const uint32_t *__restrict__ count, const uint32_t *__restrict__ offset,
                                                     const uint32_t *__restrict__ list, uint32_t cap, uint32_t *__restrict__ n_cand,
                                                     unsigned long long *__restrict__ cand) {
    const int lane = threadIdx.x;
    const uint32_t g = blockIdx.x;
    const uint32_t leaf = group_leaf[g];
    const uint32_t k0 = (g - group_first[leaf]) * 64u;  // position of this group inside the leaf's list
    const uint32_t have = count[leaf] - k0;             // >= 1 by construction
    const bool live = (uint32_t)lane < have;
    const uint32_t q = list[offset[leaf] + k0 + (live ? lane : 0)];
    f2v qv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qv[k] = reinterpret_cast<const f2v *>(qry)[(size_t)q * 8 + k];
    const float thr = live ? __uint_as_float(tau[q]) : __uint_as_float(0xFF800000u);  // -Inf: nothing passes
    const f2v *p = reinterpret_cast<const f2v *>(tree.pts) + (size_t)leaf * tree.leaf_size * 8;
    const uint32_t *ix = tree.index + (size_t)leaf * tree.leaf_size;
    for (uint32_t j = 0; j < tree.leaf_size; j += kKnnUnroll) {
        float dist[kKnnUnroll];
#pragma unroll
        for (int u = 0; u < kKnnUnroll; ++u) dist[u] = knn_pair(qv, p + (size_t)(j + u) * 8);
#pragma unroll
        for (int u = 0; u < kKnnUnroll; ++u) {
            const bool pass = dist[u] <= thr;  // false for NaN
            if (__any(pass)) {
                if (pass) {
                    const uint32_t slot = atomicAdd(n_cand + q, 1u);
                    if (slot < cap) cand[(size_t)q * cap + slot] = ((unsigned long long)__float_as_uint(dist[u]) << 32) | ix[j + u];
                }
            }
        }
    }
}

// One wave per query: the 16 smallest candidate keys, ascending; a key is distance bits << 32 | index and
// distances are non-negative, so key order is (distance, index) order.  More candidates than slots: flagged.
__global__ __launch_bounds__(64) void k_knn_select(const uint32_t *__restrict__ tau, const uint32_t *__restrict__ n_cand,
                                                  const unsigned long long *__restrict__ cand, uint32_t cap, uint32_t n_qry, float *__restrict__ out_d, uint32_t *__restrict__ out_i,
                                                  uint32_t *__restrict__ overflow) {
    constexpr unsigned long long kNone = 0x7F800000FFFFFFFFull;
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    if (tau[q] == 0xFF800000u) return;  // settled in an earlier round
    const uint32_t total = n_cand[q];
    const uint32_t n = total < cap ? total : cap;
    if (total > cap && lane == 0) overflow[q] = 1u;
    unsigned long long mine = kNone;  // smallest key not yet taken among this lane's share (keys lane, lane + 64, ...)
    auto refill = [&](unsigned long long above, bool first) {
        unsigned long long best = kNone;
        for (uint32_t k = lane; k < n; k += 64) {
            const unsigned long long v = cand[(size_t)q * cap + k];
            if ((first || v > above) && v < best) best = v;
        }
        return best;
    };
    mine = refill(0ull, true);
    for (int t = 0; t < kTopK; ++t) {
        unsigned long long w = mine;
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(w, off);
            w = (o < w) ? o : w;
        }
        if (lane == 0) {
            out_d[(size_t)q * kTopK + t] = __uint_as_float((uint32_t)(w >> 32));
            out_i[(size_t)q * kTopK + t] = (uint32_t)w;
        }
        if (mine == w && w != kNone) mine = refill(w, false);  // keys are unique (one per point), so exactly one lane owns w
    }
}

__global__ __launch_bounds__(256) void k_knn_groups(const uint32_t *__restrict__ group_first, const uint32_t *__restrict__ count,
                                                   uint32_t n_leaves, uint32_t *__restrict__ group_leaf) {
    const uint32_t leaf = blockIdx.x * 256u + threadIdx.x;
    if (leaf >= n_leaves) return;
    const uint32_t g0 = group_first[leaf], g1 = g0 + (count[leaf] + 63u) / 64u;
    for (uint32_t g = g0; g < g1; ++g) group_leaf[g] = leaf;
}

hipError_t launch_knn_groups(const uint32_t *group_first, const uint32_t *count, uint32_t n_leaves, uint32_t *group_leaf, hipStream_t stream) {
    hipLaunchKernelGGL(k_knn_groups, dim3((n_leaves + 255) / 256), dim3(256), 0, stream, group_first, count, n_leaves, group_leaf);
    return hipGetLastError();
}

hipError_t launch_knn_pairs(const KnnTreeView &t, const float *qry, const uint32_t *subset, uint32_t n_qry, const uint32_t *tau,
                            uint32_t *count, const uint32_t *offset, uint32_t *list, unsigned long long *masks, hipStream_t stream) {
    if (n_qry == 0) return hipSuccess;
    KnnTreeArgs tree{t.pts, t.index, t.leaf_lo, t.leaf_hi, t.n_leaves, t.leaf_size, 0, 0};
    uint32_t ysplit = 1;
    while (ysplit * 2 <= t.n_leaves && ysplit < 16) ysplit *= 2;  // more workgroups for small query sets
    hipLaunchKernelGGL(k_knn_pairs, dim3((n_qry + 255) / 256, ysplit), dim3(256), 0, stream, tree, qry, subset, n_qry, tau, count, offset, list,
                       masks);
    return hipGetLastError();
}

hipError_t launch_knn_leaf_scan(const KnnTreeView &t, const float *qry, const uint32_t *tau, uint32_t n_groups, const uint32_t *group_leaf,
                                const uint32_t *group_first, const uint32_t *count, const uint32_t *offset, const uint32_t *list,
                                uint32_t cap, uint32_t *n_cand, unsigned long long *cand, hipStream_t stream) {
    if (n_groups == 0) return hipSuccess;
    KnnTreeArgs tree{t.pts, t.index, t.leaf_lo, t.leaf_hi, t.n_leaves, t.leaf_size, 0, 0};
    hipLaunchKernelGGL(k_knn_leaf_scan, dim3(n_groups), dim3(64), 0, stream, tree, qry, tau, group_leaf, group_first, count, offset, list, cap,
                       n_cand, cand);
    return hipGetLastError();
}

// After a round: a query that filled its slots continues with the 16th distance among the candidates it did
// collect (real points, so still an upper bound, and tighter); every other query is settled (tau = -Inf lists it
// nowhere).  status[0] counts the queries that continue, status[1] those whose bound did not move.
__global__ __launch_bounds__(256) void k_knn_retau(uint32_t *__restrict__ tau, const uint32_t *__restrict__ overflow, const float *__restrict__ out_d,
                                                  uint32_t n_qry, uint32_t *__restrict__ status) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= n_qry) return;
    if (tau[q] == 0xFF800000u) return;
    if (!overflow[q]) { tau[q] = 0xFF800000u; return; }
    const uint32_t t = __float_as_uint(out_d[(size_t)q * kTopK + kTopK - 1]);
    atomicAdd(&status[0], 1u);
    if (t >= tau[q]) atomicAdd(&status[1], 1u);  // bit patterns of non-negative floats order like the floats
    else tau[q] = t;
}

// The bound phase runs the wave walk in several splits for parallelism; each split's list holds the 16 best of
// its own leaves.  The 16th smallest distance over all splits' lists is the bound the union supports.
__global__ __launch_bounds__(256) void k_knn_tau_merge(const float *__restrict__ lists, uint32_t splits, uint32_t n_qry,
                                                      uint32_t *__restrict__ tau) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= n_qry) return;
    float best[kTopK];
#pragma unroll
    for (int t = 0; t < kTopK; ++t) best[t] = lists[(size_t)q * kTopK + t];  // split 0, ascending
    for (uint32_t sp = 1; sp < splits; ++sp) {
        for (int t = 0; t < kTopK; ++t) {
            float c = lists[((size_t)sp * n_qry + q) * kTopK + t];
            if (!(c < best[kTopK - 1])) break;  // ascending: nothing further in this list can enter
#pragma unroll
            for (int u = 0; u < kTopK; ++u) {
                const bool lt = c < best[u];
                const float keep = lt ? c : best[u];
                c = lt ? best[u] : c;
                best[u] = keep;
            }
        }
    }
    tau[q] = __float_as_uint(best[kTopK - 1]);
}

hipError_t launch_knn_tau_merge(const float *lists, uint32_t splits, uint32_t n_qry, uint32_t *tau, hipStream_t stream) {
    hipLaunchKernelGGL(k_knn_tau_merge, dim3((n_qry + 255) / 256), dim3(256), 0, stream, lists, splits, n_qry, tau);
    return hipGetLastError();
}

hipError_t launch_knn_select(const uint32_t *tau, const uint32_t *n_cand, const unsigned long long *cand, uint32_t cap, uint32_t n_qry,
                             float *out_d, uint32_t *out_i, uint32_t *overflow, hipStream_t stream) {
    if (n_qry == 0) return hipSuccess;
    hipLaunchKernelGGL(k_knn_select, dim3(n_qry), dim3(64), 0, stream, tau, n_cand, cand, cap, n_qry, out_d, out_i, overflow);
    return hipGetLastError();
}

hipError_t launch_knn_retau(uint32_t *tau, const uint32_t *overflow, const float *out_d, uint32_t n_qry, uint32_t *status, hipStream_t stream) {
    hipLaunchKernelGGL(k_knn_retau, dim3((n_qry + 255) / 256), dim3(256), 0, stream, tau, overflow, out_d, n_qry, status);
    return hipGetLastError();
}

hipError_t launch_knn_euclid16(const float *pts, uint32_t n_pts, const float *qry, uint32_t n_qry, uint32_t idx_base,
                               uint32_t splits, bool angular, float *out_d, uint32_t *out_i, hipStream_t stream) {
    if (n_qry == 0 || splits == 0) return hipSuccess;
    const uint32_t per = (n_pts + splits - 1) / splits;
    dim3 grid((n_qry + 255) / 256, splits);
    if (angular) hipLaunchKernelGGL(k_knn_euclid16<true>, grid, dim3(256), 0, stream, pts, n_pts, qry, n_qry, idx_base, per, out_d, out_i);
    else hipLaunchKernelGGL(k_knn_euclid16<false>, grid, dim3(256), 0, stream, pts, n_pts, qry, n_qry, idx_base, per, out_d, out_i);
    return hipGetLastError();
}

}  // namespace rtdp
