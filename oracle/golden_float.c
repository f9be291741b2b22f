/*
 * TEST INFRASTRUCTURE ONLY -- oracle "B": scalar native-float restatement of
 * the datapath's four operations.  Never linked into, loaded by or called from
 * the product; bench.py times it as the CPU baseline ("port").
 *
 * PARITY STATUS: see hardfloat_rtl.c.  B must agree bit-for-bit with oracle A
 * (tests/test_oracle_ab.py); A follows the RTL gate by gate, B lets the host
 * FPU do the IEEE-754 binary32 round-to-nearest-even adds and multiplies.
 *
 * Build requirements (oracle/Makefile): -ffp-contract=off, no -ffast-math, SSE2
 * scalar math (no x87 excess precision), FTZ/DAZ left off.
 *
 * Semantics restated (reference file:line under src/main/scala/raytracer_datapath/):
 *   a (+) b, a (-) b, a (x) b : one RNE rounding each, subnormals kept; a NaN
 *       result is the canonical 0x7FC00000 (external/hardfloat/.../RoundAnyRawFNToRecFN.scala:249,277,280).
 *       Only NaNs that reach an output are canonicalised here: NaN payloads are
 *       invisible to add/mul/compare, and the selects of the box path can never
 *       deliver a NaN to tmin_out because a hit needs lt(tmin,tmax) to be true.
 *   ge(p,q) = gt || eq of CompareRecFN (CompareRecFN.scala:54-81): false when
 *       either is NaN, -0 == +0.  C's >= on floats is exactly that predicate.
 *   M(p,q) = ge(p,q) ? p : q     m(p,q) = ge(p,q) ? q : p
 *       (RecFNCompareSelect.scala:41-50 through CompareAndSwapRecFN.scala:39-46)
 */
#include "dp_bundle.h"
#include <string.h>

static inline float f_of(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t u_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t canon(float f) { return (f != f) ? 0x7FC00000u : u_of(f); }
static inline float canonf(float f) { return f_of(canon(f)); }
static inline float fneg(float f) { return f_of(u_of(f) ^ 0x80000000u); } /* FNHelpers.scala:12-19 */

static inline float selM(float p, float q) { return (p >= q) ? p : q; }
static inline float selm(float p, float q) { return (p >= q) ? q : p; }

/* One compare-and-swap of the sort network on (key, index) pairs:
 * keep when key_i >= key_j, otherwise exchange.  QuadSortRecFNWithIndex.scala:61-82 */
#define CAS(i, j)                                                       \
    do {                                                                \
        if (!(key[i] >= key[j])) {                                      \
            float tk = key[i]; key[i] = key[j]; key[j] = tk;            \
            uint32_t ti = idx[i]; idx[i] = idx[j]; idx[j] = ti;         \
        }                                                               \
    } while (0)

/* Quadbox: UnifiedDatapath.scala:167-220 (sub), :346-398 (mul), :534-631
 * (flip, max-of-4, min-of-4, hit, miss -> +Inf), :1117-1132 (sort + reverse). */
static inline void gold_box_one(const float org[3], const float dir[3], const float inv[3], float extent,
                                const float box[4][6] /* x_min,y_min,z_min,x_max,y_max,z_max */,
                                float tmin_out[4], uint32_t isect[4], uint32_t bidx[4]) {
    float key[4];
    uint32_t hit[4], idx[4] = { 0, 1, 2, 3 };
    for (int b = 0; b < 4; ++b) {
        float tn[3], tx[3];
        for (int a = 0; a < 3; ++a) {
            float lo = (box[b][a] - org[a]) * inv[a];
            float hi = (box[b][a + 3] - org[a]) * inv[a];
            int keep = (dir[a] >= 0.0f);
            tn[a] = keep ? lo : hi;
            tx[a] = keep ? hi : lo;
        }
        float tmin = selM(selM(tn[0], tn[2]), selM(tn[1], 0.0f));
        float tmax = selm(selm(tx[0], tx[2]), selm(tx[1], extent));
        hit[b] = (tmin < tmax);
        key[b] = hit[b] ? tmin : f_of(0x7F800000u);
    }
    CAS(0, 2); CAS(1, 3); CAS(0, 1); CAS(2, 3); CAS(1, 2);
    for (int k = 0; k < 4; ++k) {
        tmin_out[k] = key[3 - k];
        bidx[k] = idx[3 - k];
        isect[k] = hit[bidx[k]];
    }
}

/* Triangle: UnifiedDatapath.scala:118-166, :293-345, :486-533, :688-726,
 * :753-781, :844-871, :898-913, :990-1005, :1087-1116. */
static inline float at3(const float v[3], uint32_t k) { return (k & 3) == 0 ? v[0] : ((k & 3) == 1 ? v[1] : v[2]); }

static inline void gold_tri_one(const float org[3], const float shear[3], uint32_t kx, uint32_t ky, uint32_t kz,
                                const float vtx[3][3] /* A,B,C */, float *t_num, float *t_denom, uint32_t *hit) {
    float P[3][3], X[3], Y[3], Z[3];
    for (int v = 0; v < 3; ++v) {
        for (int a = 0; a < 3; ++a) P[v][a] = vtx[v][a] - org[a];
        float pz = at3(P[v], kz);
        float sx = fneg(shear[0]) * pz;
        float sy = fneg(shear[1]) * pz;
        Z[v] = shear[2] * pz;
        X[v] = at3(P[v], kx) + sx;
        Y[v] = at3(P[v], ky) + sy;
    }
    enum { A = 0, B = 1, C = 2 };
    float U = X[C] * Y[B] - Y[C] * X[B];
    float V = X[A] * Y[C] - Y[A] * X[C];
    float W = X[B] * Y[A] - Y[B] * X[A];
    float den = (U + V) + W;
    float num = (U * Z[A] + V * Z[B]) + W * Z[C];
    *hit = !((U < 0.0f) || (V < 0.0f) || (W < 0.0f) || (num < 0.0f)) && !(den == 0.0f);
    *t_num = canonf(num);
    *t_denom = canonf(den);
}

/* Euclidean beat: UnifiedDatapath.scala:221-245, :399-419, :632-649, :783-803,
 * :915-935, :1007-1027; the accumulate of :1133-1154 is done by the caller. */
static inline float gold_euclid_beat(const float a[16], const float b[16], uint32_t mask) {
    float s[16];
    for (int i = 0; i < 16; ++i) {
        float d = ((mask >> i) & 1) ? (a[i] - b[i]) : 0.0f;
        s[i] = d * d;
    }
    for (int i = 0; i < 8; ++i) s[i] = s[i] + s[i + 8];
    for (int i = 0; i < 4; ++i) s[i] = s[i] + s[i + 4];
    for (int i = 0; i < 2; ++i) s[i] = s[i] + s[i + 2];
    return s[0] + s[1];
}

/* Angular beat: UnifiedDatapath.scala:246-268, :420-453, :650-676, :804-832,
 * :936-964; query = a[0..7], candidate = b[0..7] (IOSpec.scala:229-243). */
static inline void gold_angular_beat(const float q[8], const float c[8], uint32_t mask, float *dot, float *norm) {
    float p[8], n[8];
    for (int i = 0; i < 8; ++i) {
        float qi = ((mask >> i) & 1) ? q[i] : 0.0f;
        float ci = ((mask >> i) & 1) ? c[i] : 0.0f;
        p[i] = qi * ci;
        n[i] = ci * ci;
    }
    for (int i = 0; i < 4; ++i) { p[i] = p[i] + p[i + 4]; n[i] = n[i] + n[i + 4]; }
    for (int i = 0; i < 2; ++i) { p[i] = p[i] + p[i + 2]; n[i] = n[i] + n[i + 2]; }
    *dot = p[0] + p[1];
    *norm = n[0] + n[1];
}

/* ------------------------------------------------------------ SoA exports */

/* in[34]: org xyz, dir xyz, inv xyz, extent, then box b: x_min,y_min,z_min,x_max,y_max,z_max.
 * flags = isIntersect[3:0] | boxIndex[0] << 4 | boxIndex[1] << 6 | boxIndex[2] << 8 | boxIndex[3] << 10. */
void gold_box_soa(uint64_t n, const float *const in[34], float *const tmin[4], uint32_t *flags) {
    for (uint64_t i = 0; i < n; ++i) {
        float org[3] = { in[0][i], in[1][i], in[2][i] };
        float dir[3] = { in[3][i], in[4][i], in[5][i] };
        float inv[3] = { in[6][i], in[7][i], in[8][i] };
        float box[4][6];
        for (int b = 0; b < 4; ++b)
            for (int f = 0; f < 6; ++f) box[b][f] = in[10 + 6 * b + f][i];
        float t[4];
        uint32_t is[4], bi[4];
        gold_box_one(org, dir, inv, in[9][i], box, t, is, bi);
        uint32_t w = 0;
        for (int k = 0; k < 4; ++k) {
            tmin[k][i] = t[k];
            w |= (is[k] << k) | (bi[k] << (4 + 2 * k));
        }
        flags[i] = w;
    }
}

/* in[15]: org xyz, shear xyz, A xyz, B xyz, C xyz; k = kx | ky << 2 | kz << 4. */
void gold_tri_soa(uint64_t n, const float *const in[15], const uint32_t *k, float *t_num, float *t_denom, uint32_t *hit) {
    for (uint64_t i = 0; i < n; ++i) {
        float org[3] = { in[0][i], in[1][i], in[2][i] };
        float sh[3] = { in[3][i], in[4][i], in[5][i] };
        float vtx[3][3];
        for (int v = 0; v < 3; ++v)
            for (int a = 0; a < 3; ++a) vtx[v][a] = in[6 + 3 * v + a][i];
        gold_tri_one(org, sh, k[i] & 3, (k[i] >> 2) & 3, (k[i] >> 4) & 3, vtx, &t_num[i], &t_denom[i], &hit[i]);
    }
}

/* a, b: [n][16]; ctl = mask | reset << 16; acc_io: accumulator register in/out. */
void gold_euclid_beats(uint64_t n, const float *a, const float *b, const uint32_t *ctl, float *out, float *acc_io) {
    float acc = *acc_io;
    for (uint64_t i = 0; i < n; ++i) {
        float r = acc + gold_euclid_beat(a + 16 * i, b + 16 * i, ctl[i] & 0xFFFFu);
        out[i] = canonf(r);
        acc = ((ctl[i] >> 16) & 1) ? 0.0f : r;
    }
    *acc_io = canonf(acc);
}

/* q, c: [n][8]; ctl = mask | reset << 16. */
void gold_angular_beats(uint64_t n, const float *q, const float *c, const uint32_t *ctl, float *dot, float *norm, float *dot_io, float *norm_io) {
    float ad = *dot_io, an = *norm_io;
    for (uint64_t i = 0; i < n; ++i) {
        float p, m;
        gold_angular_beat(q + 8 * i, c + 8 * i, ctl[i] & 0xFFu, &p, &m);
        float rd = ad + p, rn = an + m;
        dot[i] = canonf(rd);
        norm[i] = canonf(rn);
        if ((ctl[i] >> 16) & 1) { ad = 0.0f; an = 0.0f; }
        else                    { ad = rd; an = rn; }
    }
    *dot_io = canonf(ad);
    *norm_io = canonf(an);
}

/* ------------------------------------------------------------ AoS export */

void gold_eval_bundles(const dp_in_bundle *in, dp_out_bundle *out, uint64_t n, int extended, dp_accum_state *state) {
    float acc = 0.0f, ad = 0.0f, an = 0.0f;
    if (state) { acc = f_of(state->euclidean); ad = f_of(state->angular_dot); an = f_of(state->angular_norm); }
    for (uint64_t i = 0; i < n; ++i) {
        const dp_in_bundle *x = &in[i];
        dp_out_bundle *o = &out[i];
        memset(o, 0, sizeof *o);
        o->opcode = x->opcode & 3;
        const float org[3] = { f_of(x->ray.origin.x), f_of(x->ray.origin.y), f_of(x->ray.origin.z) };
        const uint32_t reset = x->euclidean_reset_accum & 1;
        switch (x->opcode & 3) {
        case DP_OP_TRIANGLE: {
            const float sh[3] = { f_of(x->ray.shear.x), f_of(x->ray.shear.y), f_of(x->ray.shear.z) };
            const float vtx[3][3] = { { f_of(x->triangle.A.x), f_of(x->triangle.A.y), f_of(x->triangle.A.z) },
                                      { f_of(x->triangle.B.x), f_of(x->triangle.B.y), f_of(x->triangle.B.z) },
                                      { f_of(x->triangle.C.x), f_of(x->triangle.C.y), f_of(x->triangle.C.z) } };
            float num, den;
            gold_tri_one(org, sh, x->ray.kx, x->ray.ky, x->ray.kz, vtx, &num, &den, &o->triangle_hit);
            o->t_num = u_of(num);
            o->t_denom = u_of(den);
        } break;
        case DP_OP_QUADBOX: {
            const float dir[3] = { f_of(x->ray.dir.x), f_of(x->ray.dir.y), f_of(x->ray.dir.z) };
            const float inv[3] = { f_of(x->ray.inv.x), f_of(x->ray.inv.y), f_of(x->ray.inv.z) };
            float box[4][6], t[4];
            for (int b = 0; b < 4; ++b) {
                const dp_aabb *bb = &x->aabb[b];
                box[b][0] = f_of(bb->x_min); box[b][1] = f_of(bb->y_min); box[b][2] = f_of(bb->z_min);
                box[b][3] = f_of(bb->x_max); box[b][4] = f_of(bb->y_max); box[b][5] = f_of(bb->z_max);
            }
            gold_box_one(org, dir, inv, f_of(x->ray.extent), box, t, o->isIntersect, o->boxIndex);
            for (int k = 0; k < 4; ++k) o->tmin_out[k] = u_of(t[k]);
        } break;
        case DP_OP_EUCLIDEAN:
            if (extended) {
                float a[16], b[16];
                for (int j = 0; j < 16; ++j) { a[j] = f_of(x->euclidean_a[j]); b[j] = f_of(x->euclidean_b[j]); }
                float r = acc + gold_euclid_beat(a, b, x->euclidean_mask & 0xFFFFu);
                o->euclidean_accumulator = canon(r);
                acc = reset ? 0.0f : r;
            }
            break;
        default:
            if (extended) {
                float q[8], c[8], p, m;
                for (int j = 0; j < 8; ++j) { q[j] = f_of(x->euclidean_a[j]); c[j] = f_of(x->euclidean_b[j]); }
                gold_angular_beat(q, c, x->euclidean_mask & 0xFFu, &p, &m);
                float rd = ad + p, rn = an + m;
                o->angular_dot_product = canon(rd);
                o->angular_norm = canon(rn);
                if (reset) { ad = 0.0f; an = 0.0f; } else { ad = rd; an = rn; }
            }
            break;
        }
        if (extended) { o->euclidean_reset_accum = reset; o->angular_reset_accum = reset; }
    }
    if (state) { state->euclidean = canon(acc); state->angular_dot = canon(ad); state->angular_norm = canon(an); }
}

/* primitive exports, mirrored by rtl_* in hardfloat_rtl.c */
void gold_add_n(const uint32_t *a, const uint32_t *b, uint32_t *o, uint64_t n) { for (uint64_t i = 0; i < n; ++i) o[i] = canon(f_of(a[i]) + f_of(b[i])); }
void gold_sub_n(const uint32_t *a, const uint32_t *b, uint32_t *o, uint64_t n) { for (uint64_t i = 0; i < n; ++i) o[i] = canon(f_of(a[i]) - f_of(b[i])); }
void gold_mul_n(const uint32_t *a, const uint32_t *b, uint32_t *o, uint64_t n) { for (uint64_t i = 0; i < n; ++i) o[i] = canon(f_of(a[i]) * f_of(b[i])); }
void gold_cmp_n(const uint32_t *a, const uint32_t *b, uint32_t *o, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        float x = f_of(a[i]), y = f_of(b[i]);
        o[i] = (uint32_t)((x < y) | ((x == y) << 1) | ((x > y) << 2));
    }
}
