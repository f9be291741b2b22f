// Device functions of the datapath ops, shared by the batch kernels (kernels.hip) and
// the traversal driver (trace.hip).  One call = one RTL op, bit for bit.
//
// Numerics contract (what makes the results bit-identical to the RTL's HardFloat units;
// reference external/hardfloat/src/main/scala/AddRecFN.scala, MulRecFN.scala,
// CompareRecFN.scala, RoundAnyRawFNToRecFN.scala): every add / sub / mul is a separate
// IEEE-754 binary32 RNE VALU op (every includer is built with -ffp-contract=off and
// carries the pragma below); subnormals are kept; selects keep the first operand when
// a >= b and yield the second otherwise, also for NaN (never v_min / v_max); a NaN that
// reaches an output is rewritten to 0x7FC00000.
//
// File:line citations are relative to the reference's
// src/main/scala/raytracer_datapath/UnifiedDatapath.scala unless stated.
#pragma once
#include "kernels.h"

#pragma clang fp contract(off)

namespace rtdp {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float canon_nan(float f) { return (f != f) ? __uint_as_float(0x7FC00000u) : f; }

// M / m of the RTL's compare-and-swap: keep when p >= q (CompareAndSwapRecFN.scala:39-46).
__device__ __forceinline__ float sel_hi(float p, float q) { return (p >= q) ? p : q; }
__device__ __forceinline__ float sel_lo(float p, float q) { return (p >= q) ? q : p; }

// ------------------------------------------------------------------ Quadbox
// r[34] in BoxArgs order.  Stage 2 :167-220, stage 3 :346-398, stage 4 :534-631,
// stage 10 :1117-1132 (sort network QuadSortRecFNWithIndex.scala:27-33, reversed).
__device__ __forceinline__ void box_one(const float (&r)[kBoxFields], float (&t)[4], uint32_t &flags) {
    const float ext = r[9];
    bool keep[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) keep[a] = (r[3 + a] >= 0.0f);  // dir == -0 keeps, dir NaN swaps (:568-594)

    float key[4];
    uint32_t pay[4];  // box index | hit << 2 travels with its key through the network
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float tn[3], tx[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = (r[10 + 6 * b + a] - r[a]) * r[6 + a];
            const float hi = (r[13 + 6 * b + a] - r[a]) * r[6 + a];
            tn[a] = keep[a] ? lo : hi;
            tx[a] = keep[a] ? hi : lo;
        }
        // network element 0 of (x, y, z, +0) and element 3 of (x, y, z, extent) (:600-612)
        const float tmin = sel_hi(sel_hi(tn[0], tn[2]), sel_hi(tn[1], 0.0f));
        const float tmax = sel_lo(sel_lo(tx[0], tx[2]), sel_lo(tx[1], ext));
        const bool hit = (tmin < tmax);                         // :618-623
        key[b] = hit ? tmin : __uint_as_float(0x7F800000u);     // :626-629
        pay[b] = (uint32_t)b | ((uint32_t)hit << 2);
    }
#define RTDP_CAS(i, j)                                              \
    {                                                               \
        const bool keep_ij = (key[i] >= key[j]);                    \
        const float ki = keep_ij ? key[i] : key[j];                 \
        const float kj = keep_ij ? key[j] : key[i];                 \
        const uint32_t pi = keep_ij ? pay[i] : pay[j];              \
        const uint32_t pj = keep_ij ? pay[j] : pay[i];              \
        key[i] = ki; key[j] = kj; pay[i] = pi; pay[j] = pj;         \
    }
    RTDP_CAS(0, 2) RTDP_CAS(1, 3) RTDP_CAS(0, 1) RTDP_CAS(2, 3) RTDP_CAS(1, 2)
#undef RTDP_CAS
    uint32_t f = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // network output is largest-first; the RTL reverses it (:1120-1130)
        t[k] = key[3 - k];
        f |= ((pay[3 - k] >> 2) & 1u) << k;
        f |= (pay[3 - k] & 3u) << (4 + 2 * k);
    }
    flags = f;
}

// ----------------------------------------------------------------- Triangle
// CompareRecFN's lt against +0 (external/hardfloat/src/main/scala/CompareRecFN.scala:65-70):
// true exactly for the patterns 0x80000001 .. 0xFF800000 (negative, not -0, not NaN).
// Done on the integer image so that the four tests of stage 10 stay four
// independent predicates (a float formulation gets merged into v_min3_f32).
__device__ __forceinline__ bool lt_zero(float x) { return (__float_as_uint(x) - 0x80000001u) <= 0x7F7FFFFFu; }

__device__ __forceinline__ float at3(float x, float y, float z, uint32_t k) {  // Float3.at, IOSpec.scala:41-51
    return (k == 0u) ? x : ((k == 1u) ? y : z);
}

// r[15] in TriArgs order.  Stage 2 :118-166, 3 :293-345, 4 :486-533, 5 :688-726,
// 6 :753-781, 7 :844-871, 8 :898-913, 9 :990-1005, 10 :1087-1116.
__device__ __forceinline__ void tri_one(const float (&r)[kTriFields], uint32_t kw, float &t_num, float &t_denom, uint32_t &hit) {
    const uint32_t kx = kw & 3u, ky = (kw >> 2) & 3u, kz = (kw >> 4) & 3u;
    const float nsx = -r[3], nsy = -r[4], sz = r[5];  // sign flip of shear.x / shear.y is a bit flip (FNHelpers.scala:12-19)
    float X[3], Y[3], Z[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const float px = r[6 + 3 * v] - r[0];
        const float py = r[7 + 3 * v] - r[1];
        const float pz = r[8 + 3 * v] - r[2];
        const float pk = at3(px, py, pz, kz);
        const float mx = nsx * pk;
        const float my = nsy * pk;
        Z[v] = sz * pk;
        X[v] = at3(px, py, pz, kx) + mx;
        Y[v] = at3(px, py, pz, ky) + my;
    }
    // A = 0, B = 1, C = 2
    const float U = X[2] * Y[1] - Y[2] * X[1];
    const float V = X[0] * Y[2] - Y[0] * X[2];
    const float W = X[1] * Y[0] - Y[1] * X[0];
    const float uaz = U * Z[0], vbz = V * Z[1], wcz = W * Z[2];
    const float den = (U + V) + W;
    const float num = (uaz + vbz) + wcz;
    // a NaN compares false everywhere, so it hits unless t_denom == +-0
    hit = (!(lt_zero(U) || lt_zero(V) || lt_zero(W) || lt_zero(num)) && !(den == 0.0f)) ? 1u : 0u;
    t_num = canon_nan(num);
    t_denom = canon_nan(den);
}

// ------------------------------------------------------------ ray derivation
// The ray fields that the reference's software model derives from dir before a record is
// poked into the RTL (reference src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:36-82):
//   inv = 1.0f / dir                         (IEEE-754 correctly rounded division)
//   kz  = index of the largest |dir| component, ties keep the lower axis (strict >, :36-53)
//   kx  = (kz + 1) % 3, ky = (kx + 1) % 3, exchanged when dir[kz] < 0     (:64-76)
//   shear = (dir[kx] / dir[kz], dir[ky] / dir[kz], 1.0f / dir[kz])         (:78-82)
// NaN results are written as 0x7FC00000; every consumer in the datapath is blind to NaN
// sign and payload.  k packs kx | ky << 2 | kz << 4.
__device__ __forceinline__ void ray_derive(const float (&d)[3], float (&inv)[3], uint32_t &k, float (&shear)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) inv[c] = canon_nan(1.0f / d[c]);
    const float ax = fabsf(d[0]), ay = fabsf(d[1]), az = fabsf(d[2]);
    uint32_t kz = 0u;
    float m = ax;
    if (ay > m) { m = ay; kz = 1u; }
    if (az > m) { kz = 2u; }
    uint32_t kx = (kz + 1u == 3u) ? 0u : kz + 1u;
    uint32_t ky = (kx + 1u == 3u) ? 0u : kx + 1u;
    const float dz = at3(d[0], d[1], d[2], kz);
    if (dz < 0.0f) { const uint32_t t = kx; kx = ky; ky = t; }
    k = kx | (ky << 2) | (kz << 4);
    shear[0] = canon_nan(at3(d[0], d[1], d[2], kx) / dz);
    shear[1] = canon_nan(at3(d[0], d[1], d[2], ky) / dz);
    shear[2] = canon_nan(1.0f / dz);
}

}  // namespace rtdp
