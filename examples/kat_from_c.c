/* The C ABI from plain C: the four known-answer cases of the datapath (one per opcode) through
 * rt_dp_eval with host arrays, then the box and triangle cases again as tiled segments.  Every intermediate value is exact in binary32, so the expected bit
 * patterns follow from the RTL's dataflow by hand (reference
 * src/main/scala/raytracer_datapath/UnifiedDatapath.scala: Quadbox :167-220, :534-631, :1117-1132;
 * triangle :118-166 ... :1087-1116; Euclidean :221-245, :632-649, :1133-1154; angular :246-268,
 * :650-676, :1028-1064; default box and triangle of the software model,
 * src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:106-119).
 *
 *   gcc -I include examples/kat_from_c.c -L raytracer_amd/lib -lrt_dp -Wl,-rpath,raytracer_amd/lib -o kat_from_c
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "rt_dp.h"

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static int expect(const char *what, uint32_t got, uint32_t want) {
    if (got == want) return 0;
    printf("MISMATCH %s: got 0x%08x want 0x%08x\n", what, got, want);
    return 1;
}

int main(void) {
    int bad = 0;
    rt_dp_op_batch in;
    rt_dp_result_batch out;
    memset(&in, 0, sizeof in);
    memset(&out, 0, sizeof out);
    in.struct_size = sizeof in;
    out.struct_size = sizeof out;

    /* Quadbox: ray (0,0,0) -> (1,1,1), extent 4000, four default boxes (all +Inf): no hit, keys tie, boxIndex = 3,2,1,0 */
    float zero = 0.0f, one = 1.0f, extent = 4000.0f, inf = INFINITY;
    float tmin[4] = {7, 7, 7, 7};
    uint32_t flags = 0xDEAD;
    for (int a = 0; a < 3; ++a) { in.box.org[a] = &zero; in.box.dir[a] = &one; in.box.inv[a] = &one; }
    in.box.extent = &extent;
    for (int b = 0; b < 4; ++b)
        for (int f = 0; f < 6; ++f) in.box.aabb[b][f] = &inf;
    for (int k = 0; k < 4; ++k) out.box.tmin_out[k] = &tmin[k];
    out.box.flags = &flags;
    in.n[RT_DP_OP_QUADBOX] = 1;

    /* Triangle: same ray (kx, ky, kz = 1, 2, 0; shear (1,1,1)), all-zero triangle: t_num = t_denom = +0, no hit */
    uint32_t k = 1u | (2u << 2) | (0u << 4), hit = 9;
    float t_num = 7, t_denom = 7;
    for (int a = 0; a < 3; ++a) { in.tri.org[a] = &zero; in.tri.shear[a] = &one; }
    for (int v = 0; v < 3; ++v)
        for (int a = 0; a < 3; ++a) in.tri.vtx[v][a] = &zero;
    in.tri.k = &k;
    out.tri.t_num = &t_num; out.tri.t_denom = &t_denom; out.tri.hit = &hit;
    in.n[RT_DP_OP_TRIANGLE] = 1;

    /* Euclidean: a = 0..15, b = 16..1; beats with reset 0, 1, 1: sums 1376, 2752, then 1376 again from +0 */
    float ea[48], eb[48], eacc[3];
    uint32_t ectl[3] = {RT_DP_CTL(0xFFFFu, 0u), RT_DP_CTL(0xFFFFu, 1u), RT_DP_CTL(0xFFFFu, 1u)};
    for (int i = 0; i < 48; ++i) { ea[i] = (float)(i % 16); eb[i] = (float)(16 - i % 16); }
    in.euc.a = ea; in.euc.b = eb; in.euc.ctl = ectl;
    out.euc.accumulator = eacc;
    in.n[RT_DP_OP_EUCLIDEAN] = 3;

    /* Angular: query 0..7, candidate 16..9, mask 0xFF, reset 1: dot 308, norm 1292 */
    float aq[8], ac[8], dot, norm;
    uint32_t actl = RT_DP_CTL(0xFFu, 1u);
    for (int i = 0; i < 8; ++i) { aq[i] = (float)i; ac[i] = (float)(16 - i); }
    in.ang.a = aq; in.ang.b = ac; in.ang.ctl = &actl;
    out.ang.dot_product = &dot; out.ang.norm = &norm;
    in.n[RT_DP_OP_ANGULAR] = 1;

    int rc = rt_dp_eval(&in, &out);
    if (rc != RT_DP_OK) {
        printf("rt_dp_eval failed: %d (%s)\n", rc, rt_dp_last_error());
        return 2;
    }
    for (int i = 0; i < 4; ++i) bad += expect("tmin_out", bits(tmin[i]), 0x7F800000u);
    bad += expect("box flags (isIntersect 0000, boxIndex 3,2,1,0)", flags, (3u << 4) | (2u << 6) | (1u << 8) | (0u << 10));
    bad += expect("t_num", bits(t_num), 0u);
    bad += expect("t_denom", bits(t_denom), 0u);
    bad += expect("triangle_hit", hit, 0u);
    bad += expect("euclidean beat 0", bits(eacc[0]), 0x44AC0000u);
    bad += expect("euclidean beat 1", bits(eacc[1]), 0x452C0000u);
    bad += expect("euclidean beat 2", bits(eacc[2]), 0x44AC0000u);
    bad += expect("angular dot", bits(dot), 0x439A0000u);
    bad += expect("angular norm", bits(norm), 0x44A18000u);

    /* The same Quadbox and triangle cases as tiled segments (one buffer per segment, [tile][word][RT_DP_TILE_OPS]), three ops each:
     * op 1 is the case above, ops 0 and 2 are a box the ray enters at t = 2 in child 1 only, resp. zero padding. */
    enum { N = 3 };
    static float box_in[RT_DP_TILE_OPS * RT_DP_BOX_TILE_IN_WORDS], tri_in[RT_DP_TILE_OPS * RT_DP_TRI_TILE_IN_WORDS];
    static uint32_t box_out[RT_DP_TILE_OPS * RT_DP_BOX_TILE_OUT_WORDS], tri_out[RT_DP_TILE_OPS * RT_DP_TRI_TILE_OUT_WORDS];
    memset(box_out, 0x5A, sizeof box_out);
    memset(tri_out, 0x5A, sizeof tri_out);
    for (int i = 0; i < N; ++i) {
        for (int w = 0; w < 3; ++w) {                                                   /* org, dir, inv */
            box_in[RT_DP_TILED_INDEX(i, w, RT_DP_BOX_TILE_IN_WORDS)] = 0.0f;
            box_in[RT_DP_TILED_INDEX(i, 3 + w, RT_DP_BOX_TILE_IN_WORDS)] = 1.0f;
            box_in[RT_DP_TILED_INDEX(i, 6 + w, RT_DP_BOX_TILE_IN_WORDS)] = 1.0f;
        }
        box_in[RT_DP_TILED_INDEX(i, 9, RT_DP_BOX_TILE_IN_WORDS)] = 4000.0f;             /* extent */
        for (int w = 10; w < 34; ++w) box_in[RT_DP_TILED_INDEX(i, w, RT_DP_BOX_TILE_IN_WORDS)] = INFINITY;
        for (int w = 0; w < 3; ++w) tri_in[RT_DP_TILED_INDEX(i, w, RT_DP_TRI_TILE_IN_WORDS)] = 0.0f;
        for (int w = 3; w < 6; ++w) tri_in[RT_DP_TILED_INDEX(i, w, RT_DP_TRI_TILE_IN_WORDS)] = 1.0f;
        for (int w = 6; w < 15; ++w) tri_in[RT_DP_TILED_INDEX(i, w, RT_DP_TRI_TILE_IN_WORDS)] = 0.0f;
        memcpy(&tri_in[RT_DP_TILED_INDEX(i, 15, RT_DP_TRI_TILE_IN_WORDS)], &k, 4);
    }
    /* op 0: child 1 = [2, 3]^3: tmin = 2, tmax = 3, the only hit.  Keys (Inf, 2, Inf, Inf) through the network (0,2) (1,3) (0,1) (2,3)
     * (1,2), larger key to the lower position, ties stay: only (1,3) swaps, giving children 0, 3, 2, 1 largest first; the output is that
     * reversed: 1, 2, 3, 0 (UnifiedDatapath.scala:1117-1132) */
    for (int f = 0; f < 3; ++f) {
        box_in[RT_DP_TILED_INDEX(0, 10 + 6 * 1 + f, RT_DP_BOX_TILE_IN_WORDS)] = 2.0f;
        box_in[RT_DP_TILED_INDEX(0, 10 + 6 * 1 + 3 + f, RT_DP_BOX_TILE_IN_WORDS)] = 3.0f;
    }
    rt_dp_op_batch tin;
    rt_dp_result_batch tout;
    memset(&tin, 0, sizeof tin);
    memset(&tout, 0, sizeof tout);
    tin.struct_size = sizeof tin;
    tout.struct_size = sizeof tout;
    tin.flags = RT_DP_F_BOX_TILED | RT_DP_F_TRI_TILED;
    tin.n[RT_DP_OP_QUADBOX] = N;
    tin.n[RT_DP_OP_TRIANGLE] = N;
    tin.box.org[0] = box_in;
    tout.box.tmin_out[0] = (float *)box_out;
    tin.tri.org[0] = tri_in;
    tout.tri.t_num = (float *)tri_out;
    rc = rt_dp_eval(&tin, &tout);
    if (rc != RT_DP_OK) {
        printf("rt_dp_eval (tiled) failed: %d (%s)\n", rc, rt_dp_last_error());
        return 2;
    }
    for (int i = 1; i < N; ++i) {
        for (int w = 0; w < 4; ++w) bad += expect("tiled tmin_out", box_out[RT_DP_TILED_INDEX(i, w, RT_DP_BOX_TILE_OUT_WORDS)], 0x7F800000u);
        bad += expect("tiled box flags", box_out[RT_DP_TILED_INDEX(i, 4, RT_DP_BOX_TILE_OUT_WORDS)], (3u << 4) | (2u << 6) | (1u << 8) | (0u << 10));
    }
    bad += expect("tiled tmin_out[0] of the hit", box_out[RT_DP_TILED_INDEX(0, 0, RT_DP_BOX_TILE_OUT_WORDS)], 0x40000000u);
    for (int w = 1; w < 4; ++w) bad += expect("tiled tmin_out of the misses", box_out[RT_DP_TILED_INDEX(0, w, RT_DP_BOX_TILE_OUT_WORDS)], 0x7F800000u);
    bad += expect("tiled box flags of the hit (isIntersect 0001, boxIndex 1,2,3,0)", box_out[RT_DP_TILED_INDEX(0, 4, RT_DP_BOX_TILE_OUT_WORDS)],
                  1u | (1u << 4) | (2u << 6) | (3u << 8) | (0u << 10));
    for (int i = 0; i < N; ++i) {
        bad += expect("tiled t_num", tri_out[RT_DP_TILED_INDEX(i, 0, RT_DP_TRI_TILE_OUT_WORDS)], 0u);
        bad += expect("tiled t_denom", tri_out[RT_DP_TILED_INDEX(i, 1, RT_DP_TRI_TILE_OUT_WORDS)], 0u);
        bad += expect("tiled triangle_hit", tri_out[RT_DP_TILED_INDEX(i, 2, RT_DP_TRI_TILE_OUT_WORDS)], 0u);
    }
    bad += expect("word behind op N - 1 of the tiled result buffer", box_out[RT_DP_TILED_INDEX(N, 0, RT_DP_BOX_TILE_OUT_WORDS)], 0x5A5A5A5Au);
    printf(bad ? "KAT FAILED\n" : "KAT OK (%s)\n", rt_dp_version());
    return bad ? 1 : 0;
}
