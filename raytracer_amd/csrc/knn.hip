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

__device__ __forceinline__ float knn_canon(float f) { return (f != f) ? __uint_as_float(0x7FC00000u) : f; }

__global__ __launch_bounds__(256) void k_knn_euclid16(const float *__restrict__ pts, uint32_t n_pts, const float *__restrict__ qry,
                                                     uint32_t n_qry, uint32_t idx_base, uint32_t pts_per_split,
                                                     float *__restrict__ out_d, uint32_t *__restrict__ out_i) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    const bool live = q < n_qry;
    const uint32_t qq = live ? q : (n_qry - 1);
    float qv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) qv[i] = qry[(size_t)qq * 16 + i];

    float bd[kTopK];
    uint32_t bi[kTopK];
#pragma unroll
    for (int j = 0; j < kTopK; ++j) { bd[j] = __uint_as_float(0x7F800000u); bi[j] = 0xFFFFFFFFu; }

    const uint32_t j0 = blockIdx.y * pts_per_split;
    const uint32_t j1 = (j0 + pts_per_split < n_pts) ? j0 + pts_per_split : n_pts;
    for (uint32_t j = j0; j < j1; ++j) {
        const float *p = pts + (size_t)j * 16;  // wave-uniform address
        float s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float d = qv[i] - p[i];
            s[i] = d * d;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = s[i] + s[i + 8];
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = s[i] + s[i + 4];
#pragma unroll
        for (int i = 0; i < 2; ++i) s[i] = s[i] + s[i + 2];
        const float dist = knn_canon(0.0f + (s[0] + s[1]));
        const uint32_t idx = idx_base + j;
        // beats the current worst entry?  (lt, or eq with a smaller index)
        const bool better = (dist < bd[kTopK - 1]) || (dist == bd[kTopK - 1] && idx < bi[kTopK - 1]);
        if (__any(better)) {
            float cd = better ? dist : __uint_as_float(0x7FC00000u);  // a NaN candidate changes nothing below
            uint32_t ci = idx;
#pragma unroll
            for (int t = 0; t < kTopK; ++t) {  // one pass of insertion into the ascending list
                const bool lt = (cd < bd[t]) || (cd == bd[t] && ci < bi[t]);
                const float nd = lt ? cd : bd[t];
                const uint32_t ni = lt ? ci : bi[t];
                cd = lt ? bd[t] : cd;
                ci = lt ? bi[t] : ci;
                bd[t] = nd;
                bi[t] = ni;
            }
        }
    }
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
