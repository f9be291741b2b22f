/*
 * TEST INFRASTRUCTURE ONLY -- oracle "A": integer-level restatement of the
 * reference RTL.  Never linked into, loaded by or called from the product.
 *
 * PARITY STATUS: the reference ships no bit-level golden vectors for the
 * datapath (its top-level test is unseeded and tolerance-based), and its
 * Chisel/Verilator tool-chain is absent here, so this restatement is pinned by
 * (1) the reference's only fixed vectors, the FNFlipSign sign-flip KATs,
 * (2) the properties its own tests assert, re-checked in tests/, and
 * (3) bit-for-bit agreement with an independent native-float model (oracle "B")
 *     whose adds/muls are executed by the host FPU.
 * Euclidean/angular results are only printed by the reference's tests, never
 * asserted: for those two opcodes bit-level parity is UNPINNED by the reference.
 *
 * Every FP value below is a 33-bit HardFloat "recoded" word held in a uint64_t,
 * exactly as it travels between the RTL's pipeline stages.  The arithmetic units
 * are bit-vector transliterations (same intermediate widths, same truncations)
 * of the vendored HardFloat sources, specialised to expWidth=8, sigWidth=24,
 * round_near_even, tininess_beforeRounding (UnifiedDatapath.scala:73-74).
 */
#include "dp_bundle.h"
#include <string.h>

typedef uint64_t u64;
typedef int64_t i64;

#define MASK(n) ((n) >= 64 ? ~0ULL : ((1ULL << (n)) - 1ULL))
#define BITS(v, hi, lo) (((u64)(v) >> (lo)) & MASK((hi) - (lo) + 1))
#define BIT(v, i) (((u64)(v) >> (i)) & 1ULL)

/* RawFloat(expWidth=8, sigWidth=w): external/hardfloat/src/main/scala/common.scala:68-77 */
typedef struct raw_float {
    int isNaN, isInf, isZero, sign;
    i64 sExp; /* SInt(10.W) */
    u64 sig;  /* UInt((w+1).W) */
} raw_float;

/* sign-extend the low n bits */
static i64 sext(u64 v, int n) {
    v &= MASK(n);
    if (v & (1ULL << (n - 1))) return (i64)(v | ~MASK(n));
    return (i64)v;
}

/* countLeadingZeros over a w-bit vector = PriorityEncoder(in.asBools.reverse);
 * an all-zero input selects the last index (w-1).  primitives.scala:89-92 */
static unsigned clz_w(u64 v, int w) {
    for (int i = 0; i < w; ++i)
        if (BIT(v, w - 1 - i)) return (unsigned)i;
    return (unsigned)(w - 1);
}

/* lowMask(in, topBound, bottomBound), primitives.scala:45-85.  Closed form of
 * the recursive shifter: result bit j (0 <= j < |top-bottom|) is
 *   top > bottom :  in > bottom + j
 *   top < bottom :  in < bottom - j
 * (tests/test_oracle_primitives.py checks this against a literal restatement
 * of the recursion for every input value). */
static u64 low_mask(u64 in, i64 top, i64 bottom) {
    u64 r = 0;
    if (top > bottom) {
        for (i64 j = 0; j < top - bottom; ++j)
            if ((i64)in > bottom + j) r |= 1ULL << j;
    } else {
        for (i64 j = 0; j < bottom - top; ++j)
            if ((i64)in < bottom - j) r |= 1ULL << j;
    }
    return r;
}

/* orReduceBy2 / orReduceBy4, primitives.scala:96-126 */
static u64 or_reduce_by(u64 v, int w, int by) {
    int rw = (w + by - 1) / by;
    u64 r = 0;
    for (int ix = 0; ix < rw - 1; ++ix)
        if (BITS(v, ix * by + by - 1, ix * by)) r |= 1ULL << ix;
    if (BITS(v, w - 1, (rw - 1) * by)) r |= 1ULL << (rw - 1);
    return r;
}

/* rawFloatFromFN(8, 24, in): rawFloatFromFN.scala:42-72 */
static raw_float raw_from_fn(uint32_t in) {
    raw_float o;
    u64 sign = BIT(in, 31);
    u64 expIn = BITS(in, 30, 23);
    u64 fractIn = BITS(in, 22, 0);
    int isZeroExpIn = (expIn == 0);
    int isZeroFractIn = (fractIn == 0);
    u64 normDist = clz_w(fractIn, 23);                           /* 5 bits */
    u64 subnormFract = BITS(fractIn << normDist, 21, 0) << 1;    /* 23 bits */
    u64 adjustedExp =
        ((isZeroExpIn ? (normDist ^ 0x1FFULL) : expIn) + (0x80ULL | (isZeroExpIn ? 2ULL : 1ULL))) & MASK(9);
    int isZero = isZeroExpIn && isZeroFractIn;
    int isSpecial = (BITS(adjustedExp, 8, 7) == 3);
    o.isNaN = isSpecial && !isZeroFractIn;
    o.isInf = isSpecial && isZeroFractIn;
    o.isZero = isZero;
    o.sign = (int)sign;
    o.sExp = (i64)BITS(adjustedExp, 8, 0); /* zext */
    o.sig = ((u64)(!isZero) << 23) | (isZeroExpIn ? subnormFract : fractIn);
    return o;
}

/* recFNFromFN(8, 24, in): recFNFromFN.scala:42-52 */
static u64 rec_from_fn(uint32_t in) {
    raw_float r = raw_from_fn(in);
    u64 top3 = (r.isZero ? 0ULL : BITS((u64)r.sExp, 8, 6)) | (r.isNaN ? 1ULL : 0ULL);
    return ((u64)r.sign << 32) | (top3 << 29) | (BITS((u64)r.sExp, 5, 0) << 23) | BITS(r.sig, 22, 0);
}

/* rawFloatFromRecFN(8, 24, in): rawFloatFromRecFN.scala:47-63 */
static raw_float raw_from_rec(u64 in) {
    raw_float o;
    u64 exp = BITS(in, 31, 23); /* 9 bits */
    int isZero = (BITS(exp, 8, 6) == 0);
    int isSpecial = (BITS(exp, 8, 7) == 3);
    o.isNaN = isSpecial && BIT(exp, 6);
    o.isInf = isSpecial && !BIT(exp, 6);
    o.isZero = isZero;
    o.sign = (int)BIT(in, 32);
    o.sExp = (i64)exp;
    o.sig = ((u64)(!isZero) << 23) | BITS(in, 22, 0);
    return o;
}

/* fNFromRecFN(8, 24, in): fNFromRecFN.scala:43-67 */
static uint32_t fn_from_rec(u64 in) {
    raw_float r = raw_from_rec(in);
    int isSubnormal = (r.sExp < 130);
    u64 denormShiftDist = (1ULL - BITS((u64)r.sExp, 4, 0)) & MASK(5);
    u64 denormFract = BITS((r.sig >> 1) >> denormShiftDist, 22, 0);
    u64 expOut = (isSubnormal ? 0ULL : ((BITS((u64)r.sExp, 7, 0) - 129ULL) & MASK(8))) |
                 ((r.isNaN || r.isInf) ? 0xFFULL : 0ULL);
    u64 fractOut = isSubnormal ? denormFract : (r.isInf ? 0ULL : BITS(r.sig, 22, 0));
    return (uint32_t)(((u64)r.sign << 31) | (expOut << 23) | fractOut);
}

static int is_sig_nan(const raw_float *r, int sigWidth) { return r->isNaN && !BIT(r->sig, sigWidth - 2); }

/* RoundAnyRawFNToRecFN(8, 26, 8, 24, options=0) as instantiated through
 * RoundRawFNToRecFN by AddRecFN / MulRecFN; round_near_even only.
 * RoundAnyRawFNToRecFN.scala:47-288 */
static u64 round_raw_to_rec(int invalidExc, const raw_float *in /* sig is 27 bits */) {
    const i64 outMinNormExp = 130, outMinNonzeroExp = 107;
    i64 sAdjustedExp = in->sExp;
    u64 adjustedSig = in->sig & MASK(27);
    u64 doShiftSigDown1 = BIT(adjustedSig, 26);

    u64 roundMask = ((low_mask(BITS((u64)sAdjustedExp, 8, 0), outMinNormExp - 24 - 1, outMinNormExp) | doShiftSigDown1) << 2) | 3ULL; /* 27 bits */
    u64 shiftedRoundMask = roundMask >> 1;
    u64 roundPosMask = ~shiftedRoundMask & roundMask & MASK(27);
    int roundPosBit = (adjustedSig & roundPosMask) != 0;
    int anyRoundExtra = (adjustedSig & shiftedRoundMask) != 0;
    int roundIncr = roundPosBit; /* round_near_even; roundMagUp is false */
    u64 roundedSig;
    if (roundIncr) {
        u64 t = (((adjustedSig | roundMask) >> 2) + 1ULL) & MASK(26);
        u64 clr = (roundPosBit && !anyRoundExtra) ? (roundMask >> 1) : 0ULL;
        roundedSig = t & ~clr & MASK(26);
    } else {
        roundedSig = (adjustedSig & ~roundMask & MASK(27)) >> 2;
    }
    i64 sRoundedExp = sAdjustedExp + (i64)(roundedSig >> 24);
    u64 common_expOut = BITS((u64)sRoundedExp, 8, 0);
    u64 common_fractOut = doShiftSigDown1 ? BITS(roundedSig, 23, 1) : BITS(roundedSig, 22, 0);
    int common_overflow = ((sRoundedExp >> 7) >= 3);
    int common_totalUnderflow = (sRoundedExp < outMinNonzeroExp);

    int isNaNOut = invalidExc || in->isNaN;
    int notNaN_isSpecialInfOut = in->isInf;
    int commonCase = !isNaNOut && !notNaN_isSpecialInfOut && !in->isZero;
    int overflow = commonCase && common_overflow;
    int notNaN_isInfOut = notNaN_isSpecialInfOut || overflow; /* RNE rounds overflow to Inf */

    u64 signOut = isNaNOut ? 0ULL : (u64)in->sign;
    u64 expOut = (common_expOut & ~((in->isZero || common_totalUnderflow) ? 0x1C0ULL : 0ULL) &
                  ~(notNaN_isInfOut ? 0x040ULL : 0ULL) & MASK(9)) |
                 (notNaN_isInfOut ? 0x180ULL : 0ULL) | (isNaNOut ? 0x1C0ULL : 0ULL);
    u64 fractOut = (isNaNOut || in->isZero || common_totalUnderflow) ? (isNaNOut ? (1ULL << 22) : 0ULL) : common_fractOut;
    return (signOut << 32) | (expOut << 23) | fractOut;
}

/* AddRecFN(8,24): AddRecFN.scala:47-165 (AddRawFN + RoundRawFNToRecFN) */
static u64 add_rec(u64 a_rec, u64 b_rec, int subOp) {
    const int sigWidth = 24, alignDistWidth = 5;
    raw_float a = raw_from_rec(a_rec), b = raw_from_rec(b_rec);

    int effSignB = b.sign ^ subOp;
    int eqSigns = (a.sign == effSignB);
    int notEqSigns_signZero = 0; /* roundingMode != round_min */
    i64 sDiffExps = a.sExp - b.sExp;
    u64 modNatAlignDist = BITS((u64)(sDiffExps < 0 ? b.sExp - a.sExp : sDiffExps), alignDistWidth - 1, 0);
    i64 hi = sDiffExps >> alignDistWidth; /* arithmetic */
    int isMaxAlign = (hi != 0) && ((hi != -1) || (BITS((u64)sDiffExps, alignDistWidth - 1, 0) == 0));
    u64 alignDist = isMaxAlign ? MASK(alignDistWidth) : modNatAlignDist;
    int closeSubMags = !eqSigns && !isMaxAlign && (modNatAlignDist <= 1);

    /* close path */
    u64 close_alignedSigA = ((0 <= sDiffExps && BIT((u64)sDiffExps, 0)) ? (a.sig << 2) : 0ULL) |
                            ((0 <= sDiffExps && !BIT((u64)sDiffExps, 0)) ? (a.sig << 1) : 0ULL) |
                            ((sDiffExps < 0) ? a.sig : 0ULL);
    i64 close_sSigSum = sext(close_alignedSigA, 27) - sext(b.sig << 1, 26);
    u64 close_sigSum = BITS((u64)(close_sSigSum < 0 ? -close_sSigSum : close_sSigSum), sigWidth + 1, 0);
    u64 close_adjustedSigSum = close_sigSum << (sigWidth & 1);
    u64 close_reduced2SigSum = or_reduce_by(close_adjustedSigSum, 26, 2); /* 13 bits */
    u64 close_normDistReduced2 = clz_w(close_reduced2SigSum, 13);
    u64 close_nearNormDist = BITS(close_normDistReduced2 << 1, alignDistWidth - 1, 0);
    u64 close_sigOut = BITS((close_sigSum << close_nearNormDist) << 1, sigWidth + 2, 0);
    int close_totalCancellation = !BITS(close_sigOut, sigWidth + 2, sigWidth + 1);
    int close_notTotalCancellation_signOut = a.sign ^ (close_sSigSum < 0);

    /* far path */
    int far_signOut = (sDiffExps < 0) ? effSignB : a.sign;
    u64 far_sigLarger = BITS((sDiffExps < 0) ? b.sig : a.sig, sigWidth - 1, 0);
    u64 far_sigSmaller = BITS((sDiffExps < 0) ? a.sig : b.sig, sigWidth - 1, 0);
    u64 far_mainAlignedSigSmaller = (far_sigSmaller << 5) >> alignDist;
    u64 far_reduced4SigSmaller = or_reduce_by(far_sigSmaller << 2, 26, 4); /* 7 bits */
    u64 far_roundExtraMask = low_mask(BITS(alignDist, alignDistWidth - 1, 2), (sigWidth + 5) / 4, 0);
    u64 far_alignedSigSmaller = ((far_mainAlignedSigSmaller >> 3) << 1) |
                                ((BITS(far_mainAlignedSigSmaller, 2, 0) != 0) || ((far_reduced4SigSmaller & far_roundExtraMask) != 0)); /* 27 bits */
    int far_subMags = !eqSigns;
    u64 far_negAlignedSigSmaller = far_subMags ? ((1ULL << 27) | (~far_alignedSigSmaller & MASK(27))) : far_alignedSigSmaller; /* 28 bits */
    u64 far_sigSum = ((far_sigLarger << 3) + far_negAlignedSigSmaller + (u64)far_subMags) & MASK(28);
    u64 far_sigOut = BITS(far_subMags ? far_sigSum : ((far_sigSum >> 1) | BIT(far_sigSum, 0)), sigWidth + 2, 0);

    int notSigNaN_invalidExc = a.isInf && b.isInf && !eqSigns;
    int notNaN_isInfOut = a.isInf || b.isInf;
    int addZeros = a.isZero && b.isZero;
    int notNaN_specialCase = notNaN_isInfOut || addZeros;
    int notNaN_isZeroOut = addZeros || (!notNaN_isInfOut && closeSubMags && close_totalCancellation);
    int notNaN_signOut = (eqSigns && a.sign) || (a.isInf && a.sign) || (b.isInf && effSignB) ||
                         (notNaN_isZeroOut && !eqSigns && notEqSigns_signZero) ||
                         (!notNaN_specialCase && closeSubMags && !close_totalCancellation && close_notTotalCancellation_signOut) ||
                         (!notNaN_specialCase && !closeSubMags && far_signOut);
    i64 common_sExpOut = ((closeSubMags || sDiffExps < 0) ? b.sExp : a.sExp) -
                         (i64)(closeSubMags ? close_nearNormDist : (u64)far_subMags);
    u64 common_sigOut = closeSubMags ? close_sigOut : far_sigOut;

    raw_float o;
    int invalidExc = is_sig_nan(&a, sigWidth) || is_sig_nan(&b, sigWidth) || notSigNaN_invalidExc;
    o.isInf = notNaN_isInfOut;
    o.isZero = notNaN_isZeroOut;
    o.sExp = sext((u64)common_sExpOut, 10);
    o.isNaN = a.isNaN || b.isNaN;
    o.sign = notNaN_signOut;
    o.sig = common_sigOut;
    return round_raw_to_rec(invalidExc, &o);
}

/* MulRecFN(8,24): MulRecFN.scala:47-129 (MulFullRawFN + MulRawFN + round) */
static u64 mul_rec(u64 a_rec, u64 b_rec) {
    const int sigWidth = 24;
    raw_float a = raw_from_rec(a_rec), b = raw_from_rec(b_rec);
    int notSigNaN_invalidExc = (a.isInf && b.isZero) || (a.isZero && b.isInf);
    raw_float o;
    o.isInf = a.isInf || b.isInf;
    o.isZero = a.isZero || b.isZero;
    o.sign = a.sign ^ b.sign;
    o.sExp = sext((u64)(a.sExp + b.sExp - 256), 10);
    u64 full = BITS(a.sig * b.sig, sigWidth * 2 - 1, 0);
    o.sig = ((full >> (sigWidth - 2)) << 1) | (BITS(full, sigWidth - 3, 0) != 0); /* 27 bits */
    o.isNaN = a.isNaN || b.isNaN;
    int invalidExc = is_sig_nan(&a, sigWidth) || is_sig_nan(&b, sigWidth) || notSigNaN_invalidExc;
    return round_raw_to_rec(invalidExc, &o);
}

/* CompareRecFN(8,24): CompareRecFN.scala:42-82 */
typedef struct cmp_out { int lt, eq, gt; } cmp_out;
static cmp_out cmp_rec(u64 a_rec, u64 b_rec) {
    raw_float a = raw_from_rec(a_rec), b = raw_from_rec(b_rec);
    int ordered = !a.isNaN && !b.isNaN;
    int bothInfs = a.isInf && b.isInf;
    int bothZeros = a.isZero && b.isZero;
    int eqExps = (a.sExp == b.sExp);
    int common_ltMags = (a.sExp < b.sExp) || (eqExps && (a.sig < b.sig));
    int common_eqMags = eqExps && (a.sig == b.sig);
    int ordered_lt = !bothZeros && ((a.sign && !b.sign) ||
                                    (!bothInfs && ((a.sign && !common_ltMags && !common_eqMags) || (!b.sign && common_ltMags))));
    int ordered_eq = bothZeros || ((a.sign == b.sign) && (bothInfs || common_eqMags));
    cmp_out o;
    o.lt = ordered && ordered_lt;
    o.eq = ordered && ordered_eq;
    o.gt = ordered && !ordered_lt && !ordered_eq;
    return o;
}

/* FNFlipSign on a 33-bit recFN word: FNHelpers.scala:12-19 */
static u64 flip_sign_rec(u64 v) { return v ^ (1ULL << 32); }

/* RecFNCompareSelect: RecFNCompareSelect.scala:36-50.  c/d carry an arbitrary
 * payload (35 bits when an index rides along). */
static void compare_select(int option, u64 a, u64 b, u64 c, u64 d, u64 *c_out, u64 *d_out) {
    cmp_out r = cmp_rec(a, b);
    if (r.gt || (option && r.eq)) { *c_out = c; *d_out = d; }
    else                          { *c_out = d; *d_out = c; }
}

static const int CAS_SEQ[5][2] = { {0, 2}, {1, 3}, {0, 1}, {2, 3}, {1, 2} };

/* QuadSortRecFN: QuadSortRecFN.scala:30-86 (CAS = CompareAndSwapRecFN.scala:32-46) */
static void quad_sort(const u64 in[4], u64 out[4]) {
    u64 e[4];
    memcpy(e, in, sizeof e);
    for (int s = 0; s < 5; ++s) {
        int i = CAS_SEQ[s][0], j = CAS_SEQ[s][1];
        u64 x, y;
        compare_select(1, e[i], e[j], e[i], e[j], &x, &y);
        e[i] = x; e[j] = y;
    }
    memcpy(out, e, sizeof e);
}

/* QuadSortRecFNWithIndex: QuadSortRecFNWithIndex.scala:27-95 */
static void quad_sort_index(const u64 in[4], u64 idx_out[4]) {
    u64 e[4], ix[4] = { 0, 1, 2, 3 };
    memcpy(e, in, sizeof e);
    for (int s = 0; s < 5; ++s) {
        int i = CAS_SEQ[s][0], j = CAS_SEQ[s][1];
        u64 c = (ix[i] << 33) | e[i], d = (ix[j] << 33) | e[j], co, dn;
        compare_select(1, e[i], e[j], c, d, &co, &dn);
        ix[i] = BITS(co, 34, 33); e[i] = BITS(co, 32, 0);
        ix[j] = BITS(dn, 34, 33); e[j] = BITS(dn, 32, 0);
    }
    memcpy(idx_out, ix, sizeof ix);
}

/* Float3.at(idx): IOSpec.scala:41-51 -- 0 -> x, 1 -> y, anything else -> z */
typedef struct rf3 { u64 x, y, z; } rf3;
static u64 at3(const rf3 *v, uint32_t k) { return (k & 3) == 0 ? v->x : ((k & 3) == 1 ? v->y : v->z); }
static rf3 rec3(const dp_f3 *v) { rf3 r = { rec_from_fn(v->x), rec_from_fn(v->y), rec_from_fn(v->z) }; return r; }

#define REC_ZERO (rec_from_fn(0x00000000u))     /* DatapathConstants._zero_RecFN       UnifiedDatapath.scala:26 */
#define REC_POS_INF (rec_from_fn(0x7f800000u))  /* DatapathConstants._positive_inf_RecFN UnifiedDatapath.scala:30 */

/* The datapath's three accumulator registers, in recoded form. */
typedef struct rtl_state { u64 euclid, ang_dot, ang_norm; } rtl_state;

/* One record through stage 1 .. stage 10 and the output conversion.
 * extended == 0 models RaytracerParams(support_euclidean = None): opcodes 2/3
 * fall through every stage untouched (UnifiedDatapath.scala:224,247) and the
 * euclidean/angular output fields do not exist (reported as 0). */
static void rtl_step(const dp_in_bundle *in, rtl_state *st, int extended, dp_out_bundle *out) {
    memset(out, 0, sizeof *out);
    const uint32_t opcode = in->opcode & 3;
    out->opcode = opcode;

    /* stage 1: FN -> recFN on every operand (UnifiedDatapath.scala:1200-1225) */
    rf3 origin = rec3(&in->ray.origin), dir = rec3(&in->ray.dir), inv = rec3(&in->ray.inv), shear = rec3(&in->ray.shear);
    u64 extent = rec_from_fn(in->ray.extent);
    const uint32_t kx = in->ray.kx & 3, ky = in->ray.ky & 3, kz = in->ray.kz & 3;
    u64 vec_a[16], vec_b[16];
    for (int i = 0; i < 16; ++i) { vec_a[i] = rec_from_fn(in->euclidean_a[i]); vec_b[i] = rec_from_fn(in->euclidean_b[i]); }
    const uint32_t mask = in->euclidean_mask & 0xFFFFu;
    const int reset = in->euclidean_reset_accum & 1;

    /* pipeline-bundle fields that reach the output; all power up as 0 (:1205) */
    u64 tmin[4] = { 0, 0, 0, 0 }, t_num = 0, t_denom = 0, vec_accum_val = 0, ang_dot_val = 0, ang_norm_val = 0;
    uint32_t isIntersect[4] = { 0, 0, 0, 0 }, boxIndex[4] = { 0, 0, 0, 0 }, triangle_hit = 0;

    if (opcode == DP_OP_TRIANGLE) {
        rf3 tA = rec3(&in->triangle.A), tB = rec3(&in->triangle.B), tC = rec3(&in->triangle.C);
        /* stage 2 (:118-166): translate the vertices */
        rf3 *T[3] = { &tA, &tB, &tC };
        for (int v = 0; v < 3; ++v) {
            T[v]->x = add_rec(T[v]->x, origin.x, 1);
            T[v]->y = add_rec(T[v]->y, origin.y, 1);
            T[v]->z = add_rec(T[v]->z, origin.z, 1);
        }
        /* stage 3 (:293-345): shear products on the kz component */
        rf3 S[3];
        for (int v = 0; v < 3; ++v) {
            u64 pz = at3(T[v], kz);
            S[v].x = mul_rec(flip_sign_rec(shear.x), pz);
            S[v].y = mul_rec(flip_sign_rec(shear.y), pz);
            S[v].z = mul_rec(shear.z, pz);
        }
        /* stage 4 (:486-533) */
        for (int v = 0; v < 3; ++v) {
            S[v].x = add_rec(at3(T[v], kx), S[v].x, 0);
            S[v].y = add_rec(at3(T[v], ky), S[v].y, 0);
        }
        const rf3 A = S[0], B = S[1], C = S[2];
        /* stage 5 (:688-726) */
        u64 U = mul_rec(C.x, B.y), V = mul_rec(A.x, C.y), W = mul_rec(B.x, A.y);
        u64 Us = mul_rec(C.y, B.x), Vs = mul_rec(A.y, C.x), Ws = mul_rec(B.y, A.x);
        /* stage 6 (:753-781) */
        U = add_rec(U, Us, 1); V = add_rec(V, Vs, 1); W = add_rec(W, Ws, 1);
        /* stage 7 (:844-871) */
        u64 U_Az = mul_rec(U, A.z), V_Bz = mul_rec(V, B.z), W_Cz = mul_rec(W, C.z);
        /* stage 8 (:898-913) */
        t_denom = add_rec(U, V, 0);
        t_num = add_rec(U_Az, V_Bz, 0);
        /* stage 9 (:990-1005) */
        t_denom = add_rec(t_denom, W, 0);
        t_num = add_rec(t_num, W_Cz, 0);
        /* stage 10 (:1087-1116) */
        cmp_out cU = cmp_rec(U, REC_ZERO), cV = cmp_rec(V, REC_ZERO), cW = cmp_rec(W, REC_ZERO);
        cmp_out cD = cmp_rec(t_denom, REC_ZERO), cN = cmp_rec(t_num, REC_ZERO);
        if (cU.lt || cV.lt || cW.lt || cN.lt) triangle_hit = 0;
        else if (cD.eq) triangle_hit = 0;
        else triangle_hit = 1;
    } else if (opcode == DP_OP_QUADBOX) {
        u64 key[4];
        uint32_t hit[4];
        for (int b = 0; b < 4; ++b) {
            const dp_aabb *bb = &in->aabb[b];
            /* stage 2 (:167-220) */
            u64 lo[3] = { add_rec(rec_from_fn(bb->x_min), origin.x, 1), add_rec(rec_from_fn(bb->y_min), origin.y, 1), add_rec(rec_from_fn(bb->z_min), origin.z, 1) };
            u64 hi[3] = { add_rec(rec_from_fn(bb->x_max), origin.x, 1), add_rec(rec_from_fn(bb->y_max), origin.y, 1), add_rec(rec_from_fn(bb->z_max), origin.z, 1) };
            /* stage 3 (:346-398) */
            const u64 iv[3] = { inv.x, inv.y, inv.z }, dv[3] = { dir.x, dir.y, dir.z };
            u64 tn[3], tx[3];
            for (int a = 0; a < 3; ++a) {
                lo[a] = mul_rec(lo[a], iv[a]);
                hi[a] = mul_rec(hi[a], iv[a]);
                /* stage 4 (:537-594): keep when dir >= +0, otherwise swap */
                compare_select(1, dv[a], REC_ZERO, lo[a], hi[a], &tn[a], &tx[a]);
            }
            /* stage 4 (:600-612) */
            u64 s_in[4] = { tn[0], tn[1], tn[2], REC_ZERO }, s_out[4];
            quad_sort(s_in, s_out);
            u64 tmin_i = s_out[0];
            u64 l_in[4] = { tx[0], tx[1], tx[2], extent }, l_out[4];
            quad_sort(l_in, l_out);
            u64 tmax_i = l_out[3];
            hit[b] = (uint32_t)cmp_rec(tmin_i, tmax_i).lt;      /* :618-623 */
            key[b] = hit[b] ? tmin_i : REC_POS_INF;             /* :626-629 */
        }
        /* stage 10 (:1117-1132) */
        u64 sorted[4], sidx[4];
        quad_sort_index(key, sidx);
        quad_sort(key, sorted);
        for (int k = 0; k < 4; ++k) {
            boxIndex[k] = (uint32_t)sidx[3 - k];
            tmin[k] = sorted[3 - k];
            isIntersect[k] = hit[boxIndex[k]];
        }
    } else if (opcode == DP_OP_EUCLIDEAN) {
        if (extended) {
            u64 s[16];
            for (int i = 0; i < 16; ++i) {
                u64 d = ((mask >> i) & 1) ? add_rec(vec_a[i], vec_b[i], 1) : REC_ZERO; /* stage 2 (:221-245) */
                s[i] = mul_rec(d, d);                                                    /* stage 3 (:399-419) */
            }
            for (int i = 0; i < 8; ++i) s[i] = add_rec(s[i], s[i + 8], 0); /* stage 4 (:632-649) */
            for (int i = 0; i < 4; ++i) s[i] = add_rec(s[i], s[i + 4], 0); /* stage 6 (:783-803) */
            for (int i = 0; i < 2; ++i) s[i] = add_rec(s[i], s[i + 2], 0); /* stage 8 (:915-935) */
            s[0] = add_rec(s[0], s[1], 0);                                 /* stage 9 (:1007-1027) */
            vec_accum_val = add_rec(st->euclid, s[0], 0);                  /* stage 10 (:1133-1154) */
            st->euclid = reset ? 0 : vec_accum_val;
        }
    } else { /* DP_OP_ANGULAR */
        if (extended) {
            u64 p[8], n[8];
            for (int i = 0; i < 8; ++i) {
                /* stage 2 (:246-268): query = vec_a[0..7], candidate = vec_b[0..7] (IOSpec.scala:229-243) */
                u64 q = ((mask >> i) & 1) ? vec_a[i] : REC_ZERO;
                u64 c = ((mask >> i) & 1) ? vec_b[i] : REC_ZERO;
                p[i] = mul_rec(q, c); /* stage 3 (:420-453) */
                n[i] = mul_rec(c, c);
            }
            for (int i = 0; i < 4; ++i) { p[i] = add_rec(p[i], p[i + 4], 0); n[i] = add_rec(n[i], n[i + 4], 0); } /* stage 4 (:650-676) */
            for (int i = 0; i < 2; ++i) { p[i] = add_rec(p[i], p[i + 2], 0); n[i] = add_rec(n[i], n[i + 2], 0); } /* stage 6 (:804-832) */
            p[0] = add_rec(p[0], p[1], 0); n[0] = add_rec(n[0], n[1], 0);                                         /* stage 8 (:936-964) */
            ang_dot_val = add_rec(st->ang_dot, p[0], 0);                                                          /* stage 9 (:1028-1064) */
            ang_norm_val = add_rec(st->ang_norm, n[0], 0);
            if (reset) { st->ang_dot = 0; st->ang_norm = 0; }
            else       { st->ang_dot = ang_dot_val; st->ang_norm = ang_norm_val; }
        }
    }

    /* output conversion (UnifiedDatapath.scala:1239-1271) */
    for (int k = 0; k < 4; ++k) {
        out->tmin_out[k] = fn_from_rec(tmin[k]);
        out->isIntersect[k] = isIntersect[k];
        out->boxIndex[k] = boxIndex[k];
    }
    out->t_denom = fn_from_rec(t_denom);
    out->t_num = fn_from_rec(t_num);
    out->triangle_hit = triangle_hit;
    if (extended) {
        out->euclidean_accumulator = fn_from_rec(vec_accum_val);
        out->euclidean_reset_accum = (uint32_t)reset;
        out->angular_dot_product = fn_from_rec(ang_dot_val);
        out->angular_norm = fn_from_rec(ang_norm_val);
        out->angular_reset_accum = (uint32_t)reset;
    }
}

/* ------------------------------------------------------------------ exports */

/* Evaluate n records in stream order.  state carries the accumulators in and
 * out as IEEE bit patterns (NULL = power-up +0, result discarded). */
void rtl_eval_bundles(const dp_in_bundle *in, dp_out_bundle *out, uint64_t n, int extended, dp_accum_state *state) {
    rtl_state st = { 0, 0, 0 };
    if (state) {
        st.euclid = rec_from_fn(state->euclidean);
        st.ang_dot = rec_from_fn(state->angular_dot);
        st.ang_norm = rec_from_fn(state->angular_norm);
    }
    for (uint64_t i = 0; i < n; ++i) rtl_step(&in[i], &st, extended, &out[i]);
    if (state) {
        state->euclidean = fn_from_rec(st.euclid);
        state->angular_dot = fn_from_rec(st.ang_dot);
        state->angular_norm = fn_from_rec(st.ang_norm);
    }
}

/* Primitive-level entry points, used to compare each unit against the host FPU. */
uint32_t rtl_add(uint32_t a, uint32_t b) { return fn_from_rec(add_rec(rec_from_fn(a), rec_from_fn(b), 0)); }
uint32_t rtl_sub(uint32_t a, uint32_t b) { return fn_from_rec(add_rec(rec_from_fn(a), rec_from_fn(b), 1)); }
uint32_t rtl_mul(uint32_t a, uint32_t b) { return fn_from_rec(mul_rec(rec_from_fn(a), rec_from_fn(b))); }
/* bit0 = lt, bit1 = eq, bit2 = gt */
uint32_t rtl_cmp(uint32_t a, uint32_t b) { cmp_out r = cmp_rec(rec_from_fn(a), rec_from_fn(b)); return (uint32_t)(r.lt | (r.eq << 1) | (r.gt << 2)); }
uint32_t rtl_roundtrip(uint32_t a) { return fn_from_rec(rec_from_fn(a)); }
uint64_t rtl_rec_from_fn(uint32_t a) { return rec_from_fn(a); }
uint32_t rtl_fn_from_rec(uint64_t a) { return fn_from_rec(a); }
uint32_t rtl_flip_sign_fn(uint32_t a) { return a ^ 0x80000000u; }                       /* FNFlipSign on Bits(32.W) */
uint32_t rtl_flip_sign_via_rec(uint32_t a) { return fn_from_rec(flip_sign_rec(rec_from_fn(a))); } /* ... on the recoded word, as stage 3 uses it */
uint64_t rtl_low_mask(uint64_t in, int64_t top, int64_t bottom) { return low_mask(in, top, bottom); }
void rtl_quad_sort(const uint32_t in[4], uint32_t out[4], uint32_t idx[4]) {
    u64 r[4], o[4], ix[4];
    for (int i = 0; i < 4; ++i) r[i] = rec_from_fn(in[i]);
    quad_sort(r, o);
    quad_sort_index(r, ix);
    for (int i = 0; i < 4; ++i) { out[i] = fn_from_rec(o[i]); idx[i] = (uint32_t)ix[i]; }
}
/* option in {0,1}; returns 1 when (c_out, d_out) == (c, d), 0 when swapped */
uint32_t rtl_compare_select_keeps(uint32_t option, uint32_t a, uint32_t b) {
    u64 co, dn;
    compare_select((int)option, rec_from_fn(a), rec_from_fn(b), 1, 2, &co, &dn);
    return co == 1;
}
/* bulk versions for fast randomized comparison against the FPU */
void rtl_add_n(const uint32_t *a, const uint32_t *b, uint32_t *o, uint64_t n) { for (uint64_t i = 0; i < n; ++i) o[i] = rtl_add(a[i], b[i]); }
void rtl_sub_n(const uint32_t *a, const uint32_t *b, uint32_t *o, uint64_t n) { for (uint64_t i = 0; i < n; ++i) o[i] = rtl_sub(a[i], b[i]); }
void rtl_mul_n(const uint32_t *a, const uint32_t *b, uint32_t *o, uint64_t n) { for (uint64_t i = 0; i < n; ++i) o[i] = rtl_mul(a[i], b[i]); }
void rtl_cmp_n(const uint32_t *a, const uint32_t *b, uint32_t *o, uint64_t n) { for (uint64_t i = 0; i < n; ++i) o[i] = rtl_cmp(a[i], b[i]); }
