// Record streams in HBM, fused form: EnhancedInputBundle images in arrival order are read
// ONCE, evaluated in place and written back as EnhancedOutputBundle images, without moving
// the operands into per-opcode arrays first (stream.hip does that; this file is the faster
// way to the same bits).
//
//   k_stream_eval   one wave per 64 consecutive records.  The wave copies its 21 KiB of
//                   records into LDS with coalesced 16-byte loads (a lane-per-record read
//                   would touch 64 cache lines per instruction), then each lane evaluates
//                   its own record: Quadbox and triangle ops completely; for a distance beat
//                   the tree sum(s) of the beat, which do not depend on the accumulator.
//                   The 21-word output records go back through LDS to coalesced stores.
//                   Side arrays receive, per record, tag = opcode | reset << 2 and the
//                   beat's tree sums, plus one bit per 64 records and distance opcode
//                   saying "a reset of this opcode occurs here".
//   k_stream_fold   per distance opcode: the accumulator is a strictly sequential float sum
//                   over the beats of one vector pair IN STREAM ORDER (reference
//                   src/main/scala/raytracer_datapath/UnifiedDatapath.scala:1133-1154,
//                   :1028-1064), so a workgroup owns whole pairs: its range of records runs
//                   from the first pair boundary at or after blockIdx * tile to the first one
//                   at or after (blockIdx + 1) * tile (a boundary follows a record of this
//                   opcode that carries the reset bit; found through the bit summaries, so
//                   long stretches without one cost 128 Ki records per step).  Inside a
//                   sub-tile the opcode's beats are compacted into LDS in stream order, one
//                   lane per pair adds them up, and the results are patched into the output
//                   records.
// Output fields of foreign opcodes are zero and the two *_reset_accum fields echo the input
// bit for every opcode (:1205, :1254, :1267-1268).
#include "ops.hpp"

#pragma clang fp contract(off)

namespace rtdp {

constexpr int kRecIn = 84;    // words per rt_dp_input_bundle
constexpr int kRecOut = 21;   // words per rt_dp_output_bundle
constexpr int kWaveRecs = 64;

struct StreamSide {
    uint32_t *tag;        // [n]   opcode | reset << 2 | outputs final << 3
    float *sum0, *sum1;   // [n]   tree sums of a distance beat (Euclidean: sum0; angular: dot, norm)
    uint32_t *rbits[2];   // [ceil(n / 2048)] bit b of word w: records [(32 w + b) * 64, +64) hold a reset of opcode 2 / 3
    uint32_t *flags;      // [1] opcode 2 present, [2] opcode 3 present (per launch); [8] an opcode > 3 was seen (sticky)
    uint8_t *wavebits;    // [ceil(n / 64)] per wave of k_stream_eval: 1 / 2 = a beat of opcode 2 / 3 occurs, 4 / 8 = one with its reset bit
};

__device__ __forceinline__ float tree16(const float (&s)[16]) {  // (i, i+8), (i, i+4), (i, i+2), (0, 1)
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = s[i] + s[i + 8];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = t[i] + t[i + 4];
    return (t[0] + t[2]) + (t[1] + t[3]);
}

__device__ __forceinline__ float tree8(const float (&s)[8]) {  // (i, i+4), (i, i+2), (0, 1)
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = s[i] + s[i + 4];
    return (t[0] + t[2]) + (t[1] + t[3]);
}

// One lane evaluates the record it owns (s_rec + 84 * lane): Quadbox and triangle ops completely, for a distance beat the
// tree sum(s) of the beat, which do not depend on the accumulator.  w: the 21 output words (accumulator fields still 0).
// Returns false when the opcode is not one of the four.
__device__ __forceinline__ bool stream_eval_record(const uint32_t *s_rec, int lane, bool have, int extended, uint32_t (&w)[kRecOut], uint32_t &op,
                                                   uint32_t &reset, float &sum0, float &sum1) {
    bool known = true;
    // 16-byte LDS reads of the own record: 21 is odd, so the lanes of a read spread over all banks
    const u4v *rc = reinterpret_cast<const u4v *>(s_rec + lane * kRecIn);
    const u4v c0 = rc[0];
    op = have ? c0.x : 0xFFFFFFFFu;
    reset = have ? (rc[20].w & 1u) : 0u;
#pragma unroll
    for (int k = 0; k < kRecOut; ++k) w[k] = 0u;
    sum0 = 0.0f;
    sum1 = 0.0f;
    if (have) {
        w[0] = op;
        if (extended) { w[17] = reset; w[20] = reset; }
    }
    if (op == 1u) {  // Quadbox: words 1..10 (ray) and 17..40 (four AABBs)
        uint32_t W[44];
        W[1] = c0.y; W[2] = c0.z; W[3] = c0.w;
#pragma unroll
        for (int c = 1; c <= 10; ++c) {
            if (c == 3) continue;
            const u4v v = rc[c];
            W[4 * c] = v.x; W[4 * c + 1] = v.y; W[4 * c + 2] = v.z; W[4 * c + 3] = v.w;
        }
        float r[kBoxFields], t[4];
#pragma unroll
        for (int f = 0; f < 10; ++f) r[f] = __uint_as_float(W[1 + f]);
#pragma unroll
        for (int f = 0; f < 24; ++f) r[10 + f] = __uint_as_float(W[17 + f]);
        uint32_t fl;
        box_one(r, t, fl);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w[1 + k] = __float_as_uint(t[k]);
            w[5 + k] = (fl >> k) & 1u;
            w[9 + k] = (fl >> (4 + 2 * k)) & 3u;
        }
    } else if (op == 0u) {  // Triangle: words 1..3 origin, 11..13 k, 14..16 shear, 41..49 vertices
        const u4v c2 = rc[2], c3 = rc[3], c4 = rc[4], c10 = rc[10], c11 = rc[11], c12 = rc[12];
        float r[kTriFields];
        r[0] = __uint_as_float(c0.y); r[1] = __uint_as_float(c0.z); r[2] = __uint_as_float(c0.w);
        r[3] = __uint_as_float(c3.z); r[4] = __uint_as_float(c3.w); r[5] = __uint_as_float(c4.x);
        r[6] = __uint_as_float(c10.y); r[7] = __uint_as_float(c10.z); r[8] = __uint_as_float(c10.w);
        r[9] = __uint_as_float(c11.x); r[10] = __uint_as_float(c11.y); r[11] = __uint_as_float(c11.z); r[12] = __uint_as_float(c11.w);
        r[13] = __uint_as_float(c12.x); r[14] = __uint_as_float(c12.y);
        const uint32_t kw = (c2.w & 3u) | ((c3.x & 3u) << 2) | ((c3.y & 3u) << 4);
        float num, den;
        uint32_t hit;
        tri_one(r, kw, num, den, hit);
        w[13] = __float_as_uint(num);
        w[14] = __float_as_uint(den);
        w[15] = hit;
    } else if (op == 2u) {  // Euclidean beat: a = words 50..65, b = 66..81, mask = word 82
        uint32_t W[36];  // words 48..83
#pragma unroll
        for (int c = 12; c <= 20; ++c) {
            const u4v v = rc[c];
            W[4 * (c - 12)] = v.x; W[4 * (c - 12) + 1] = v.y; W[4 * (c - 12) + 2] = v.z; W[4 * (c - 12) + 3] = v.w;
        }
        const uint32_t mask = W[34];
        float s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool on = (mask >> i) & 1u;
            const float d = on ? (__uint_as_float(W[2 + i]) - __uint_as_float(W[18 + i])) : 0.0f;  // :221-245
            s[i] = d * d;                                                                            // :399-419
        }
        sum0 = tree16(s);
    } else if (op == 3u) {  // Angular beat: query = words 50..57, candidate = 66..73
        const u4v c12 = rc[12], c13 = rc[13], c14 = rc[14], c16 = rc[16], c17 = rc[17], c18 = rc[18], c20 = rc[20];
        const uint32_t A[8] = {c12.z, c12.w, c13.x, c13.y, c13.z, c13.w, c14.x, c14.y};
        const uint32_t B[8] = {c16.z, c16.w, c17.x, c17.y, c17.z, c17.w, c18.x, c18.y};
        const uint32_t mask = c20.z;
        float s[8], m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool on = (mask >> i) & 1u;
            const float q = on ? __uint_as_float(A[i]) : 0.0f;  // :246-268
            const float c = on ? __uint_as_float(B[i]) : 0.0f;
            s[i] = q * c;                                        // :420-453
            m[i] = c * c;
        }
        sum0 = tree8(s);
        sum1 = tree8(m);
    } else if (have) {
        known = false;  // cannot occur in the RTL's 2-bit field
    }
    return known;
}


// The X beats selected by `todo` (a subset of the wave's lanes, ascending = stream order) are added onto acc one by one;
// the owning lane keeps its beat's result in res; a beat with its reset bit restarts the sum from +0 behind it.
template <bool TWO>
__device__ __forceinline__ void stream_wave_walk(unsigned long long todo, unsigned long long resets, float s0, float s1, int lane, float &acc0,
                                                 float &acc1, float &res0, float &res1) {
    while (todo) {
        const int L = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
        todo &= todo - 1;
        const bool rs = (resets >> L) & 1ull;
        const float o0 = acc0 + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s0), L));
        if (lane == L) res0 = o0;
        acc0 = rs ? 0.0f : o0;
        if (TWO) {
            const float o1 = acc1 + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s1), L));
            if (lane == L) res1 = o1;
            acc1 = rs ? 0.0f : o1;
        }
    }
}

// ---- compact staging ---------------------------------------------------------------------------------------------
// A record is 21 chunks of 16 bytes, but an opcode reads at most 11 of them (Quadbox: ray origin / dir / inv / extent, the four
// boxes and the reset word; triangle 8; Euclidean beat 9; angular beat 7).  The evaluation kernel is bound by how many waves'
// record buffers fit a CU's LDS, so the wave keeps only the live chunks: the coalesced loads fetch whole records as before
// (the lines come in full anyway), the lane that holds a record's chunk 0 posts its opcode, and then every lane files each
// chunk it holds under slot = number of live chunks of that record's opcode below it -- or drops it.  Records are 11 slots
// apart (odd: the lanes' 16-byte reads of their own record spread over all banks): 11 KiB per wave instead of 21, twice the
// resident waves.
constexpr int kLiveStride = 11;  // slots per record
constexpr uint32_t kLiveBox = 0x1007F7u;   // chunks 0 1 2, 4..10, 20
constexpr uint32_t kLiveTri = 0x101C1Du;   // chunks 0, 2 3 4, 10 11 12, 20
constexpr uint32_t kLiveEuc = 0x1FF000u;   // chunks 12..20
constexpr uint32_t kLiveAng = 0x177000u;   // chunks 12 13 14, 16 17 18, 20

__device__ __forceinline__ uint32_t live_chunks(uint32_t op) {
    return op == 1u ? kLiveBox : op == 0u ? kLiveTri : op == 2u ? kLiveEuc : op == 3u ? kLiveAng : 0u;
}

// stream_eval_record over the compact slots of the lane's record (rc = 11 slots).  Same arithmetic, same outputs.
__device__ __forceinline__ bool stream_eval_compact(const u4v *rc, uint32_t op_in, bool have, int extended, uint32_t (&w)[kRecOut], uint32_t &op,
                                                    uint32_t &reset, float &sum0, float &sum1) {
    bool known = true;
    op = have ? op_in : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < kRecOut; ++k) w[k] = 0u;
    sum0 = 0.0f;
    sum1 = 0.0f;
    reset = 0u;
    if (op == 1u) {  // slots: chunks 0 1 2 | 4..10 | 20
        uint32_t W[44];
        const u4v c0 = rc[0], c1 = rc[1], c2 = rc[2];
        W[1] = c0.y; W[2] = c0.z; W[3] = c0.w;
        W[4] = c1.x; W[5] = c1.y; W[6] = c1.z; W[7] = c1.w;
        W[8] = c2.x; W[9] = c2.y; W[10] = c2.z; W[11] = c2.w;
#pragma unroll
        for (int c = 4; c <= 10; ++c) {
            const u4v v = rc[c - 1];
            W[4 * c] = v.x; W[4 * c + 1] = v.y; W[4 * c + 2] = v.z; W[4 * c + 3] = v.w;
        }
        reset = rc[10].w & 1u;
        float r[kBoxFields], t[4];
#pragma unroll
        for (int f = 0; f < 10; ++f) r[f] = __uint_as_float(W[1 + f]);
#pragma unroll
        for (int f = 0; f < 24; ++f) r[10 + f] = __uint_as_float(W[17 + f]);
        uint32_t fl;
        box_one(r, t, fl);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w[1 + k] = __float_as_uint(t[k]);
            w[5 + k] = (fl >> k) & 1u;
            w[9 + k] = (fl >> (4 + 2 * k)) & 3u;
        }
    } else if (op == 0u) {  // slots: chunks 0 | 2 3 4 | 10 11 12 | 20
        const u4v c0 = rc[0], c2 = rc[1], c3 = rc[2], c4 = rc[3], c10 = rc[4], c11 = rc[5], c12 = rc[6];
        reset = rc[7].w & 1u;
        float r[kTriFields];
        r[0] = __uint_as_float(c0.y); r[1] = __uint_as_float(c0.z); r[2] = __uint_as_float(c0.w);
        r[3] = __uint_as_float(c3.z); r[4] = __uint_as_float(c3.w); r[5] = __uint_as_float(c4.x);
        r[6] = __uint_as_float(c10.y); r[7] = __uint_as_float(c10.z); r[8] = __uint_as_float(c10.w);
        r[9] = __uint_as_float(c11.x); r[10] = __uint_as_float(c11.y); r[11] = __uint_as_float(c11.z); r[12] = __uint_as_float(c11.w);
        r[13] = __uint_as_float(c12.x); r[14] = __uint_as_float(c12.y);
        const uint32_t kw = (c2.w & 3u) | ((c3.x & 3u) << 2) | ((c3.y & 3u) << 4);
        float num, den;
        uint32_t hit;
        tri_one(r, kw, num, den, hit);
        w[13] = __float_as_uint(num);
        w[14] = __float_as_uint(den);
        w[15] = hit;
    } else if (op == 2u) {  // slots: chunks 12..20 = words 48..83
        uint32_t W[36];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const u4v v = rc[c];
            W[4 * c] = v.x; W[4 * c + 1] = v.y; W[4 * c + 2] = v.z; W[4 * c + 3] = v.w;
        }
        reset = W[35] & 1u;
        const uint32_t mask = W[34];
        float s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool on = (mask >> i) & 1u;
            const float d = on ? (__uint_as_float(W[2 + i]) - __uint_as_float(W[18 + i])) : 0.0f;  // :221-245
            s[i] = d * d;                                                                            // :399-419
        }
        sum0 = tree16(s);
    } else if (op == 3u) {  // slots: chunks 12 13 14 | 16 17 18 | 20
        const u4v c12 = rc[0], c13 = rc[1], c14 = rc[2], c16 = rc[3], c17 = rc[4], c18 = rc[5], c20 = rc[6];
        reset = c20.w & 1u;
        const uint32_t A[8] = {c12.z, c12.w, c13.x, c13.y, c13.z, c13.w, c14.x, c14.y};
        const uint32_t B[8] = {c16.z, c16.w, c17.x, c17.y, c17.z, c17.w, c18.x, c18.y};
        const uint32_t mask = c20.z;
        float s[8], m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool on = (mask >> i) & 1u;
            const float q = on ? __uint_as_float(A[i]) : 0.0f;  // :246-268
            const float c = on ? __uint_as_float(B[i]) : 0.0f;
            s[i] = q * c;                                        // :420-453
            m[i] = c * c;
        }
        sum0 = tree8(s);
        sum1 = tree8(m);
    } else if (have) {
        known = false;  // cannot occur in the RTL's 2-bit field
    }
    if (have) {
        w[0] = op;
        if (extended) { w[17] = reset; w[20] = reset; }
    }
    return known;
}

__global__ __launch_bounds__(64) void k_stream_eval(const uint32_t *__restrict__ in, uint64_t n, int extended, uint32_t *__restrict__ out,
                                                    StreamSide side) {
    __shared__ __align__(16) uint32_t s_rec[kWaveRecs * kLiveStride * 4];  // 11 KiB of live chunks; re-used for the output records (5.25 KiB)
    __shared__ uint32_t s_op[kWaveRecs];
    const int lane = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * kWaveRecs;
    const uint32_t live = (n - r0 < (uint64_t)kWaveRecs) ? (uint32_t)(n - r0) : (uint32_t)kWaveRecs;

    // Coalesced loads of the wave's whole records (chunk g = j * 64 + lane of its 21 KiB); fetching only the chunks a record's
    // opcode reads measured 9-11 % slower on MI355X -- the lines come in full anyway.  Chunk g belongs to record g / 21; 64 = 3 * 21 + 1,
    // so from one load to the next a lane moves three records on and one chunk up (four records on where the chunk wraps).
    {
        const u4v *src = reinterpret_cast<const u4v *>(in + r0 * kRecIn);
        const uint32_t chunks = live * (kRecIn / 4);
        u4v v[kRecIn / 4];
#pragma unroll
        for (int j = 0; j < kRecIn / 4; ++j) {
            const uint32_t g = (uint32_t)j * 64u + lane;
            v[j] = (g < chunks) ? __builtin_nontemporal_load(src + g) : u4v{0u, 0u, 0u, 0u};
        }
        const uint32_t rec0 = (uint32_t)lane / 21u, chunk0 = (uint32_t)lane % 21u;
        // the lane that holds a record's first chunk posts its opcode
#pragma unroll
        for (int j = 0; j < kRecIn / 4; ++j) {
            const uint32_t wrap = (chunk0 + (uint32_t)j >= 21u) ? 1u : 0u;
            const uint32_t c = chunk0 + (uint32_t)j - 21u * wrap, r = rec0 + 3u * (uint32_t)j + wrap;
            if (c == 0u && r < (uint32_t)kWaveRecs) s_op[r] = v[j].x;
        }
        __syncthreads();
        // every chunk goes to its slot in its record's compact row, or nowhere
#pragma unroll
        for (int j = 0; j < kRecIn / 4; ++j) {
            const uint32_t wrap = (chunk0 + (uint32_t)j >= 21u) ? 1u : 0u;
            const uint32_t c = chunk0 + (uint32_t)j - 21u * wrap, r = rec0 + 3u * (uint32_t)j + wrap;
            if (r < live) {
                const uint32_t m = live_chunks(s_op[r]);
                if ((m >> c) & 1u) reinterpret_cast<u4v *>(s_rec)[r * kLiveStride + __popc(m & ((1u << c) - 1u))] = v[j];
            }
        }
    }
    __syncthreads();

    const bool have = (uint32_t)lane < live;
    uint32_t w[kRecOut], op, reset;
    float sum0, sum1;
    if (!stream_eval_compact(reinterpret_cast<const u4v *>(s_rec) + lane * kLiveStride, s_op[lane], have, extended, w, op, reset, sum0, sum1))
        atomicOr(&side.flags[8], 1u);

    // Beats behind the first reset of their opcode in this wave start from +0 whatever came before: their accumulator
    // outputs are final here and travel with the record's coalesced store.  Only the beats in front of the first reset
    // (they continue a pair from an earlier wave) are left to k_stream_fold, which then has that many fewer 4-byte
    // patches to write into finished records.
    uint32_t fin = 0u;
    if (extended) {
        const unsigned long long e2 = __ballot(op == 2u), r2 = __ballot(op == 2u && reset != 0u);
        const unsigned long long e3 = __ballot(op == 3u), r3 = __ballot(op == 3u && reset != 0u);
        if (r2) {
            const int lo = __ffsll((long long)r2) - 1;
            const unsigned long long rest = e2 & ~((lo == 63) ? ~0ull : ((2ull << lo) - 1ull));
            float a0 = 0.0f, a1 = 0.0f, res = 0.0f, none = 0.0f;
            stream_wave_walk<false>(rest, r2, sum0, 0.0f, lane, a0, a1, res, none);
            if ((rest >> lane) & 1ull) { w[16] = __float_as_uint(canon_nan(res)); fin = 1u; }
        }
        if (r3) {
            const int lo = __ffsll((long long)r3) - 1;
            const unsigned long long rest = e3 & ~((lo == 63) ? ~0ull : ((2ull << lo) - 1ull));
            float a0 = 0.0f, a1 = 0.0f, res0 = 0.0f, res1 = 0.0f;
            stream_wave_walk<true>(rest, r3, sum0, sum1, lane, a0, a1, res0, res1);
            if ((rest >> lane) & 1ull) { w[18] = __float_as_uint(canon_nan(res0)); w[19] = __float_as_uint(canon_nan(res1)); fin = 1u; }
        }
    }

    if (have) {
        side.tag[r0 + lane] = (op & 3u) | (reset << 2) | (fin << 3);
        side.sum0[r0 + lane] = sum0;
        side.sum1[r0 + lane] = sum1;
    }
    {
        const unsigned long long e = __ballot(op == 2u), a = __ballot(op == 3u);
        const unsigned long long er = __ballot(op == 2u && reset), ar = __ballot(op == 3u && reset);
        // What the fold needs to know about this wave, as one plain byte: no atomic and no device-scope read in the life of a
        // wave (an earlier form set summary bits and launch flags with atomics guarded by a device-scope load of the flag: one
        // more memory round trip per wave).  k_stream_bits turns the bytes into the summary words and the launch flags.
        if (lane == 0) side.wavebits[blockIdx.x] = (uint8_t)((e ? 1u : 0u) | (a ? 2u : 0u) | (er ? 4u : 0u) | (ar ? 8u : 0u));
    }

    __syncthreads();  // every lane has read its record: the buffer now takes the output records
#pragma unroll
    for (int k = 0; k < kRecOut; ++k) s_rec[lane * kRecOut + k] = w[k];  // stride 21 words is odd: conflict-free
    __syncthreads();
    uint32_t *dst = out + r0 * kRecOut;
    if (live == (uint32_t)kWaveRecs) {  // 1344 words = 336 chunks of 16 bytes
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int g = j * 64 + lane;
            if (g < kWaveRecs * kRecOut / 4) __builtin_nontemporal_store(reinterpret_cast<const u4v *>(s_rec)[g], reinterpret_cast<u4v *>(dst) + g);
        }
    } else {
        for (uint32_t g = lane; g < live * kRecOut; g += 64) dst[g] = s_rec[g];
    }
}

// The per-wave bytes of k_stream_eval become the reset summaries (one bit per 64 records) and the "opcode occurs" flags.
__global__ __launch_bounds__(256) void k_stream_bits(StreamSide side, uint64_t nblk) {
    const uint64_t w = (uint64_t)blockIdx.x * 256u + threadIdx.x, nwords = (nblk + 31) >> 5;
    uint32_t r0 = 0u, r1 = 0u, any = 0u;
    if (w < nwords) {
        const u4v lo = reinterpret_cast<const u4v *>(side.wavebits)[2 * w], hi = reinterpret_cast<const u4v *>(side.wavebits)[2 * w + 1];
        const uint32_t words[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const uint32_t byte = (w * 32 + b < nblk) ? ((words[b >> 2] >> (8 * (b & 3))) & 0xFFu) : 0u;  // bytes behind the last wave are stale
            any |= byte;
            r0 |= ((byte >> 2) & 1u) << b;
            r1 |= ((byte >> 3) & 1u) << b;
        }
        side.rbits[0][w] = r0;
        side.rbits[1][w] = r1;
    }
    const int any2 = __syncthreads_or((int)(any & 1u)), any3 = __syncthreads_or((int)(any & 2u));
    if (threadIdx.x == 0) {
        if (any2) atomicOr(&side.flags[1], 1u);
        if (any3) atomicOr(&side.flags[2], 1u);
    }
}

// First boundary of opcode X at or after `nominal`: the smallest p >= nominal such that p == 0 or
// record p - 1 is an X record with the reset bit; n if there is none.  Executed by one full wave.
__device__ __forceinline__ uint64_t stream_boundary(const uint32_t *tag, const uint32_t *rbits, uint32_t want, uint64_t nominal, uint64_t n,
                                                    int lane) {
    if (nominal == 0) return 0;
    uint64_t i = nominal - 1;  // the next record to examine
    const uint64_t nblk = (n + 63) >> 6, nwords = (nblk + 31) >> 5;
    for (;;) {  // every trip advances i by at least one record: terminates
        if (i >= n) return n;
        if ((i & 2047u) == 0) {  // on a summary-word boundary: 64 words = 128 Ki records per trip
            const uint64_t w0 = i >> 11;
            const uint32_t bits = (w0 + lane < nwords) ? rbits[w0 + lane] : 0u;
            const unsigned long long m = __ballot(bits != 0u);
            if (!m) { i += 64ull * 2048ull; continue; }
            const int wl = __ffsll((long long)m) - 1;
            const uint32_t first = __shfl(bits, wl);
            i = ((w0 + wl) * 32 + (__ffs((int)first) - 1)) << 6;  // the first 64-record block holding a reset of X
        }
        const uint64_t b0 = i & ~63ull, idx = b0 + lane;
        const bool p = idx >= i && idx < n && (tag[idx] & 7u) == want;
        const unsigned long long m = __ballot(p);
        if (m) return b0 + (uint64_t)(__ffsll((long long)m) - 1) + 1;
        i = b0 + 64;
    }
}

constexpr int kFoldThreads = 64;   // one wave per workgroup: the fold is a chain of dependent memory round trips,
constexpr int kFoldSub = 256;      // so many small workgroups in flight beat few large ones; records per sub-tile
constexpr int kFoldWaves = kFoldThreads / 64;
constexpr int kFoldPasses = kFoldSub / kFoldThreads;

template <bool EUC>
__device__ __forceinline__ void stream_fold(const uint32_t *__restrict__ tag, const float *__restrict__ sum0, const float *__restrict__ sum1,
                                            const uint32_t *__restrict__ rbits, uint64_t n, uint32_t tile,
                                            const float *__restrict__ state_in, float *__restrict__ state_out, uint32_t *__restrict__ out) {
    constexpr uint32_t X = EUC ? 2u : 3u;
    __shared__ float c_sum0[kFoldSub], c_sum1[EUC ? 1 : kFoldSub], c_out0[kFoldSub], c_out1[EUC ? 1 : kFoldSub];
    __shared__ uint16_t c_pos[kFoldSub];
    __shared__ uint8_t c_rs[kFoldSub];
    __shared__ uint32_t s_cnt[kFoldPasses * kFoldWaves];
    __shared__ uint64_t s_range[2];
    __shared__ float s_carry[2][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = wave; e < 2; e += kFoldWaves) {  // both ends of the range, by different waves where there are two
        const uint64_t r = stream_boundary(tag, rbits, X | 4u, (uint64_t)(blockIdx.x + e) * tile, n, lane);
        if (lane == 0) s_range[e] = r;
    }
    if (tid == 0) {  // only the workgroup that starts at record 0 can inherit a running sum
        s_carry[0][0] = (blockIdx.x == 0) ? state_in[0] : 0.0f;
        s_carry[0][1] = (!EUC && blockIdx.x == 0) ? state_in[1] : 0.0f;
    }
    __syncthreads();
    const uint64_t lo = s_range[0], hi = s_range[1];
    if (lo >= hi) return;

    int par = 0;
    for (uint64_t base = lo; base < hi; base += kFoldSub) {
        const int cnt = (hi - base < (uint64_t)kFoldSub) ? (int)(hi - base) : kFoldSub;
        // compaction of the opcode's beats, in stream order: position p = u * 256 + tid
        uint32_t tg[kFoldPasses];
        unsigned long long mk[kFoldPasses];
#pragma unroll
        for (int u = 0; u < kFoldPasses; ++u) {
            const int p = u * kFoldThreads + tid;
            tg[u] = (p < cnt) ? tag[base + p] : 0xFFu;
            mk[u] = __ballot((tg[u] & 3u) == X && tg[u] != 0xFFu);
            if (lane == 0) s_cnt[u * kFoldWaves + wave] = (uint32_t)__popcll(mk[u]);
        }
        __syncthreads();
        int m = 0;
#pragma unroll
        for (int u = 0; u < kFoldPasses; ++u) {
            uint32_t before = 0;
            for (int k = 0; k < u * kFoldWaves + wave; ++k) before += s_cnt[k];
            const int p = u * kFoldThreads + tid;
            if ((tg[u] & 3u) == X && tg[u] != 0xFFu) {
                const uint32_t k = before + (uint32_t)__popcll(mk[u] & ((1ull << lane) - 1ull));
                c_sum0[k] = sum0[base + p];
                if (!EUC) c_sum1[k] = sum1[base + p];
                c_rs[k] = (uint8_t)((tg[u] >> 2) & 3u);  // bit 0: reset, bit 1: the output is already in the record
                c_pos[k] = (uint16_t)p;
            }
        }
        for (int k = 0; k < kFoldPasses * kFoldWaves; ++k) m += (int)s_cnt[k];
        __syncthreads();
        if (m == 0) continue;  // wave-uniform: no beat of this opcode here, the carry stays where it is

        // one lane per vector pair accumulates its beats in order
        for (int t0 = tid; t0 < m; t0 += kFoldThreads) {
            const bool starts = (t0 == 0) || ((c_rs[t0 - 1] & 1u) != 0);
            if (starts) {
                float acc0 = (t0 == 0) ? s_carry[par][0] : 0.0f;
                float acc1 = (t0 == 0) ? s_carry[par][1] : 0.0f;
                int i = t0;
                bool rs;
                do {
                    const float o0 = acc0 + c_sum0[i];
                    c_out0[i] = canon_nan(o0);
                    rs = (c_rs[i] & 1u) != 0;
                    acc0 = rs ? 0.0f : o0;
                    if (!EUC) {
                        const float o1 = acc1 + c_sum1[i];
                        c_out1[i] = canon_nan(o1);
                        acc1 = rs ? 0.0f : o1;
                    }
                    ++i;
                } while (i < m && !rs);
                if (i == m) {  // exactly one lane reaches the end of the compacted beats
                    s_carry[par ^ 1][0] = acc0;
                    s_carry[par ^ 1][1] = acc1;
                }
            }
        }
        __syncthreads();
        for (int t0 = tid; t0 < m; t0 += kFoldThreads) {  // partial-line writes: without nt each one pulls its line into L2 first
            if (c_rs[t0] & 2u) continue;  // k_stream_eval wrote this beat's outputs with the record
            uint32_t *rec = out + (base + c_pos[t0]) * kRecOut;
            if (EUC) {
                __builtin_nontemporal_store(__float_as_uint(c_out0[t0]), rec + 16);
            } else {
                __builtin_nontemporal_store(__float_as_uint(c_out0[t0]), rec + 18);
                __builtin_nontemporal_store(__float_as_uint(c_out1[t0]), rec + 19);
            }
        }
        par ^= 1;
        __syncthreads();
    }
    if (hi == n && tid == 0) {  // the workgroup that owns the last record hands the registers on
        state_out[0] = canon_nan(s_carry[par][0]);
        if (!EUC) state_out[1] = canon_nan(s_carry[par][1]);
    }
}

// blockIdx.y = 0: Euclidean, 1: angular, side by side in one launch; an opcode that does not occur in
// the stream (flags written by k_stream_eval) costs one load per workgroup.  state: {euclid, dot, norm}; state_out is
// written completely: by the workgroup that owns the last record of an opcode that occurs, and as a copy of state_in
// for an opcode that does not.
__global__ __launch_bounds__(kFoldThreads) void k_stream_fold(StreamSideView v, uint64_t n, uint32_t tile, const float *__restrict__ state_in,
                                                     float *__restrict__ state_out, uint32_t *__restrict__ out) {
    if (blockIdx.y == 0) {
        if (v.flags[1]) stream_fold<true>(v.tag, v.sum0, v.sum1, v.rbits[0], n, tile, state_in, state_out, out);
        else if (blockIdx.x == 0 && threadIdx.x == 0) state_out[0] = state_in[0];
    } else {
        if (v.flags[2]) stream_fold<false>(v.tag, v.sum0, v.sum1, v.rbits[1], n, tile, state_in + 1, state_out + 1, out);
        else if (blockIdx.x == 0 && threadIdx.x == 0) { state_out[1] = state_in[1]; state_out[2] = state_in[2]; }
    }
}

// ---- the single-pass form ------------------------------------------------------------------------------------
// k_stream_chain does everything k_stream_eval and k_stream_fold do in ONE pass over the records: the output records are
// written once, complete, and no per-record side arrays exist.  A wave still owns 64 consecutive records.  Per distance
// opcode X it adds up its own X beats in stream order inside the wave (a loop over the set bits of a ballot; the running
// sum is wave-uniform), and the running sum that enters the wave comes from the waves before it through one 64-bit word
// per wave and sum, {state, float bits}, in a workspace that is zeroed before the launch:
//   state 0  not published yet
//   state 1  "transparent": the wave holds no beat of X, the sum passes through it unchanged
//   state 2  the running sum of X behind the wave's last record
// A wave that holds a reset of X knows its outgoing sum without knowing the incoming one (only the beats behind its last
// reset count) and publishes it before it looks back; a wave without a reset publishes after it has received the incoming
// sum.  Looking back, the 64 lanes examine the 64 nearest predecessors at once and skip transparent ones; in front of
// wave 0 stand the registers that enter the stream.  The arithmetic is the same strictly sequential float sum per vector
// pair in the same order as in k_stream_fold, so the bits are the same.
//
// Progress: a wave waits only for waves with a lower index, which were dispatched before it and wait, if at all, for still
// lower ones; wave 0 never waits.  The wait is bounded all the same (a lost wave must not hang the device): after
// kStreamSpins polls a give-up is counted in the header and the wave goes on with +0.
// The host sees a non-zero count when the call returns and evaluates the stream again in the two-pass form (api.cpp).
constexpr int kStreamSpins = 1 << 20;
constexpr int kStreamSpinsDebug = 1 << 12;  // with the test hook that makes one wave withhold its sums

struct StreamChain {
    unsigned long long *sticky;   // [0] give-ups, [1] an opcode > 3 was seen; cleared once per call
    uint32_t *last;               // [0], [1]: 1 + index of the last wave that published a sum of opcode 2 / 3 (0: none); per piece
    unsigned long long *word[3];  // [waves] each: Euclidean sum, angular dot, angular norm
};

__device__ __forceinline__ unsigned long long stream_word(uint32_t state, float v) {
    return ((unsigned long long)state << 32) | (unsigned long long)__float_as_uint(v);
}

// The running sum(s) of opcode X in front of wave `b`: the nearest predecessor that is not transparent, polled until it has
// published.  Executed by the whole wave (uniform control flow).
template <bool TWO>
__device__ __forceinline__ void stream_look_back(const unsigned long long *w0, const unsigned long long *w1, long long b, float in0, float in1,
                                                 int lane, unsigned long long *giveups, int spins, float &c0, float &c1) {
    long long top = b - 1;  // the nearest predecessor not yet known to be transparent
    for (int spin = 0; spin < spins; ++spin) {
        const long long idx = top - lane;
        uint32_t st = 2u;
        float v0 = in0, v1 = in1;  // in front of wave 0: the registers entering the stream
        if (idx >= 0) {
            const unsigned long long a = __hip_atomic_load(w0 + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st = (uint32_t)(a >> 32);
            v0 = __uint_as_float((uint32_t)a);
            if (TWO) {
                const unsigned long long c = __hip_atomic_load(w1 + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(c >> 32) != st) st = 0u;  // the second word is not there yet
                v1 = __uint_as_float((uint32_t)c);
            }
        }
        const unsigned long long opaque = __ballot(st != 1u);
        if (!opaque) { top -= 64; continue; }  // 64 transparent waves: look further back (lanes in front of wave 0 count as published, so this ends)
        const int L = __builtin_amdgcn_readfirstlane(__ffsll((long long)opaque) - 1);
        const uint32_t stL = (uint32_t)__builtin_amdgcn_readlane((int)st, L);
        if (stL == 2u) {
            c0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0), L));
            c1 = TWO ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v1), L)) : 0.0f;
            return;
        }
        top -= L;  // everything nearer than L is transparent; L itself has not published yet
        __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) atomicAdd(giveups, 1ull);
    c0 = 0.0f;
    c1 = 0.0f;
}

// One distance opcode of one wave: publish, look back, accumulate.  e: lanes holding a beat of X, r: those with the reset bit.
template <bool TWO>
__device__ __forceinline__ void stream_wave_fold(unsigned long long e, unsigned long long r, float s0, float s1, int lane, long long b,
                                                 unsigned long long *w0, unsigned long long *w1, const float *state_in, unsigned long long *giveups,
                                                 long long mute, float &res0, float &res1) {
    const bool speak = (b != mute);  // test hook: wave `mute` never publishes, so that its successors' bounded wait runs out
    const int spins = (mute >= 0) ? kStreamSpinsDebug : kStreamSpins;
    if (!e) {  // no beat of X here: the sum passes through
        if (lane == 0 && speak) {
            __hip_atomic_store(w0 + b, stream_word(1u, 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (TWO) __hip_atomic_store(w1 + b, stream_word(1u, 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    unsigned long long head = e;  // the beats that continue whatever runs into this wave: up to and including the first reset
    if (r) {
        const int lo = __ffsll((long long)r) - 1;
        head = e & ((lo == 63) ? ~0ull : ((2ull << lo) - 1ull));
        // everything behind the first reset starts from +0, and what is left behind the last reset is the outgoing sum
        float a0 = 0.0f, a1 = 0.0f;
        stream_wave_walk<TWO>(e & ~head, r, s0, s1, lane, a0, a1, res0, res1);
        if (lane == 0 && speak) {
            __hip_atomic_store(w0 + b, stream_word(2u, a0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (TWO) __hip_atomic_store(w1 + b, stream_word(2u, a1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float c0, c1;
    stream_look_back<TWO>(w0, w1, b, state_in[0], TWO ? state_in[1] : 0.0f, lane, giveups, spins, c0, c1);
    stream_wave_walk<TWO>(head, r, s0, s1, lane, c0, c1, res0, res1);
    if (!r && lane == 0 && speak) {  // no reset in this wave: the outgoing sum is the incoming one plus all of its beats
        __hip_atomic_store(w0 + b, stream_word(2u, c0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (TWO) __hip_atomic_store(w1 + b, stream_word(2u, c1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(64) void k_stream_chain(const uint32_t *__restrict__ in, uint64_t n, int extended, uint32_t *__restrict__ out,
                                                     StreamChain ch, const float *__restrict__ state_in, int mute) {
    __shared__ __align__(16) uint32_t s_rec[kWaveRecs * kRecIn];  // 21 KiB; re-used for the output records
    const int lane = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * kWaveRecs;
    const uint32_t live = (n - r0 < (uint64_t)kWaveRecs) ? (uint32_t)(n - r0) : (uint32_t)kWaveRecs;
    {
        const u4v *src = reinterpret_cast<const u4v *>(in + r0 * kRecIn);
        const uint32_t chunks = live * (kRecIn / 4);
        u4v v[kRecIn / 4];
#pragma unroll
        for (int j = 0; j < kRecIn / 4; ++j) {
            const uint32_t g = (uint32_t)j * 64u + lane;
            v[j] = (g < chunks) ? __builtin_nontemporal_load(src + g) : u4v{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < kRecIn / 4; ++j) reinterpret_cast<u4v *>(s_rec)[j * 64 + lane] = v[j];
    }
    __syncthreads();
    const bool have = (uint32_t)lane < live;
    uint32_t w[kRecOut], op, reset;
    float sum0, sum1;
    const bool known = stream_eval_record(s_rec, lane, have, extended, w, op, reset, sum0, sum1);
    if (__ballot(!known) && lane == 0) atomicOr(ch.sticky + 1, 1ull);

    if (extended) {
        const long long b = blockIdx.x;
        float e0 = 0.0f, unused = 0.0f, d0 = 0.0f, d1 = 0.0f;
        stream_wave_fold<false>(__ballot(op == 2u), __ballot(op == 2u && reset != 0u), sum0, 0.0f, lane, b, ch.word[0], nullptr, state_in, ch.sticky,
                                (long long)mute, e0, unused);
        stream_wave_fold<true>(__ballot(op == 3u), __ballot(op == 3u && reset != 0u), sum0, sum1, lane, b, ch.word[1], ch.word[2], state_in + 1, ch.sticky,
                               (long long)mute, d0, d1);
        if (op == 2u) w[16] = __float_as_uint(canon_nan(e0));
        if (op == 3u) { w[18] = __float_as_uint(canon_nan(d0)); w[19] = __float_as_uint(canon_nan(d1)); }
    }

    __syncthreads();  // every lane has read its record: the buffer now takes the output records
#pragma unroll
    for (int k = 0; k < kRecOut; ++k) s_rec[lane * kRecOut + k] = w[k];  // stride 21 words is odd: conflict-free
    __syncthreads();
    uint32_t *dst = out + r0 * kRecOut;
    if (live == (uint32_t)kWaveRecs) {  // 1344 words = 336 chunks of 16 bytes
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int g = j * 64 + lane;
            if (g < kWaveRecs * kRecOut / 4) __builtin_nontemporal_store(reinterpret_cast<const u4v *>(s_rec)[g], reinterpret_cast<u4v *>(dst) + g);
        }
    } else {
        for (uint32_t g = lane; g < live * kRecOut; g += 64) dst[g] = s_rec[g];
    }
}

// The registers behind the last record: the sum published by the last wave that is not transparent (the words are final
// once k_stream_chain has finished), or the entering registers when no wave held a beat of the opcode.
__global__ __launch_bounds__(1024) void k_stream_chain_last(StreamChain ch, uint32_t waves) {
    __shared__ uint32_t s_last[2];
    if (threadIdx.x < 2) s_last[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    if (i < waves) {
        if ((uint32_t)(ch.word[0][i] >> 32) == 2u) atomicMax(&s_last[0], i + 1u);
        if ((uint32_t)(ch.word[1][i] >> 32) == 2u) atomicMax(&s_last[1], i + 1u);
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_last[threadIdx.x]) atomicMax(ch.last + threadIdx.x, s_last[threadIdx.x]);
}

__global__ void k_stream_chain_state(StreamChain ch, const float *__restrict__ state_in, float *__restrict__ state_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t le = ch.last[0], la = ch.last[1];
    state_out[0] = le ? canon_nan(__uint_as_float((uint32_t)ch.word[0][le - 1u])) : state_in[0];
    state_out[1] = la ? canon_nan(__uint_as_float((uint32_t)ch.word[1][la - 1u])) : state_in[1];
    state_out[2] = la ? canon_nan(__uint_as_float((uint32_t)ch.word[2][la - 1u])) : state_in[2];
}

// Address `off` bytes into a workspace; integer arithmetic, so that a layout can also be laid over a null base to learn its size.
template <typename T> static T *ws_at(char *base, size_t off) { return reinterpret_cast<T *>(reinterpret_cast<uintptr_t>(base) + off); }

static StreamChain stream_chain_layout(char *workspace, uint64_t capacity, size_t *piece_bytes) {
    const uint64_t waves = (capacity + 63) / 64;
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    StreamChain ch;
    size_t o = 0;
    ch.sticky = ws_at<unsigned long long>(workspace, o); o += 256;
    const size_t piece0 = o;
    ch.last = ws_at<uint32_t>(workspace, o); o += 256;
    for (int k = 0; k < 3; ++k) { ch.word[k] = ws_at<unsigned long long>(workspace, o); o += up(waves * 8); }
    if (piece_bytes) *piece_bytes = o - piece0;
    return ch;
}

size_t stream_chain_workspace(uint64_t capacity) {
    size_t piece = 0;
    (void)stream_chain_layout(nullptr, capacity, &piece);
    return 256 + piece;
}

// Clears the give-up counter and the sticky "opcode > 3" word; once per call, before the first piece.
hipError_t stream_chain_begin(char *workspace, hipStream_t s) { return hipMemsetAsync(workspace, 0, 256, s); }

// One piece of n <= capacity records in one pass; state_in / state_out: {euclid, dot, norm} in front of / behind the piece.
// *sticky_out: device address of {give-ups, opcode > 3 seen} (two 64-bit words) for the caller to read back.
hipError_t stream_chain_eval(const uint32_t *in, uint64_t n, uint64_t capacity, int extended, uint32_t *out, char *workspace, const float *state_in,
                             float *state_out, const unsigned long long **sticky_out, hipStream_t s, int mute_wave) {
    size_t piece = 0;
    const StreamChain ch = stream_chain_layout(workspace, capacity, &piece);
    const uint64_t waves = (n + 63) / 64;
    if (n > capacity || waves > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (sticky_out) *sticky_out = ch.sticky;
    if (extended) {  // every word starts as "not published"
        hipError_t e = hipMemsetAsync(ch.last, 0, piece, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_stream_chain, dim3((unsigned)waves), dim3(64), 0, s, in, n, extended, out, ch, state_in, mute_wave);
    if (extended) {
        hipLaunchKernelGGL(k_stream_chain_last, dim3((unsigned)((waves + 1023) / 1024)), dim3(1024), 0, s, ch, (uint32_t)waves);
        hipLaunchKernelGGL(k_stream_chain_state, dim3(1), dim3(64), 0, s, ch, state_in, state_out);
    }
    return hipGetLastError();
}

// Workspace for pieces of at most `capacity` records: tag, sum0, sum1, the two bit arrays, 256 bytes of flags.
static StreamSide stream_fused_layout(char *workspace, uint64_t capacity, size_t *bits_bytes) {
    const uint64_t nblk = (capacity + 63) / 64, nwords = (nblk + 31) / 32;
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    StreamSide side;
    size_t o = 0;
    side.tag = ws_at<uint32_t>(workspace, o); o += up(capacity * 4);
    side.sum0 = ws_at<float>(workspace, o); o += up(capacity * 4);
    side.sum1 = ws_at<float>(workspace, o); o += up(capacity * 4);
    side.rbits[0] = ws_at<uint32_t>(workspace, o); o += up(nwords * 4);
    side.rbits[1] = ws_at<uint32_t>(workspace, o); o += up(nwords * 4);
    side.wavebits = ws_at<uint8_t>(workspace, o); o += up(nwords * 32);  // whole 32-byte groups: k_stream_bits reads them with 16-byte loads
    side.flags = ws_at<uint32_t>(workspace, o);
    if (bits_bytes) *bits_bytes = 2 * up(nwords * 4);
    return side;
}

size_t stream_fused_workspace(uint64_t capacity) {
    size_t bits = 0;
    const StreamSide side = stream_fused_layout(nullptr, capacity, &bits);
    return static_cast<size_t>(reinterpret_cast<uintptr_t>(side.flags)) + 256;
}

// One piece of n <= capacity records.  first_piece: also clears the sticky "opcode > 3" word, once per call.
hipError_t stream_fused_eval(const uint32_t *in, uint64_t n, uint64_t capacity, int extended, uint32_t *out, char *workspace,
                             StreamSideView *view, hipStream_t s, bool first_piece) {
    size_t bits = 0;
    const StreamSide side = stream_fused_layout(workspace, capacity, &bits);
    const uint64_t nblk = (n + 63) / 64;
    if (n > capacity || nblk > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(side.flags, 0, first_piece ? 256 : 32, s);  // the per-piece flags (all flags on the first piece)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_stream_eval, dim3((unsigned)nblk), dim3(64), 0, s, in, n, extended, out, side);
    hipLaunchKernelGGL(k_stream_bits, dim3((unsigned)(((nblk + 31) / 32 + 255) / 256)), dim3(256), 0, s, side, nblk);
    view->tag = side.tag; view->sum0 = side.sum0; view->sum1 = side.sum1;
    view->rbits[0] = side.rbits[0]; view->rbits[1] = side.rbits[1]; view->flags = side.flags;
    return hipGetLastError();
}

hipError_t stream_fused_fold(const StreamSideView &v, uint64_t n, const float *state_in, float *state_out, uint32_t *out, hipStream_t s) {
    const uint32_t tile = kFoldSub;
    const uint64_t blocks = (n + tile - 1) / tile;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_stream_fold, dim3((unsigned)blocks, 2), dim3(kFoldThreads), 0, s, v, n, tile, state_in, state_out, out);
    return hipGetLastError();
}

}  // namespace rtdp
