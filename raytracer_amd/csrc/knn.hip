// Exact k-nearest-neighbour search on the datapath's Euclidean distance op.
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
// the chip; partial lists are merged afterwards (merge_topk in raytracer_amd/knn.py,
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
__device__ __forceinline__ void knn_offer(float (&bd)[kTopK], uint32_t (&bi)[kTopK], float &pd, uint32_t &pi, float dist, uint32_t idx) {
    const bool pass = dist <= bd[kTopK - 1];  // one compare filters almost every candidate (false for NaN)
    if (__any(pass)) {
        if (__any(pass && (pd == pd))) {
            knn_insert(bd, bi, pd, pi);
            pd = __uint_as_float(0x7FC00000u);
        }
        pd = pass ? dist : pd;
        pi = pass ? idx : pi;
    }
}

constexpr int kKnnUnroll = 4;  // points per loop trip: their scalar loads are issued together

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

    const uint32_t j0 = blockIdx.y * pts_per_split;
    const uint32_t j1 = (j0 + pts_per_split < n_pts) ? j0 + pts_per_split : n_pts;
    uint32_t j = j0;
    for (; j + kKnnUnroll <= j1; j += kKnnUnroll) {
        const f2v *p = reinterpret_cast<const f2v *>(pts) + (size_t)j * 8;  // wave-uniform address
        float dist[kKnnUnroll];
#pragma unroll
        for (int u = 0; u < kKnnUnroll; ++u) dist[u] = knn_pair(qv, p + u * 8);
#pragma unroll
        for (int u = 0; u < kKnnUnroll; ++u) knn_offer(bd, bi, pd, pi, dist[u], idx_base + j + u);
    }
    for (; j < j1; ++j) knn_offer(bd, bi, pd, pi, knn_pair(qv, reinterpret_cast<const f2v *>(pts) + (size_t)j * 8), idx_base + j);
    knn_insert(bd, bi, pd, pi);
    if (live) {
        const size_t o = ((size_t)blockIdx.y * n_qry + q) * kTopK;
#pragma unroll
        for (int t = 0; t < kTopK; ++t) { out_d[o + t] = bd[t]; out_i[o + t] = bi[t]; }
    }
}

hipError_t launch_knn_euclid16(const float *pts, uint32_t n_pts, const float *qry, uint32_t n_qry, uint32_t idx_base,
                               uint32_t splits, float *out_d, uint32_t *out_i, hipStream_t stream) {
    if (n_qry == 0 || splits == 0) return hipSuccess;
    const uint32_t per = (n_pts + splits - 1) / splits;
    dim3 grid((n_qry + 255) / 256, splits);
    hipLaunchKernelGGL(k_knn_euclid16, grid, dim3(256), 0, stream, pts, n_pts, qry, n_qry, idx_base, per, out_d, out_i);
    return hipGetLastError();
}

}  // namespace rtdp
