// Record streams in HBM, fused form: EnhancedInputBundle images in arrival order are read
// ONCE, evaluated in place and written back as EnhancedOutputBundle images, without moving
// the operands into per-opcode arrays first (stream.hip does that; this file is the faster
// way to the same bits).
//
//   k_stream_eval   one wave per 64 consecutive records.  The wave copies its 21 KiB of
//                   records through registers into LDS with coalesced 16-byte loads (a
//                   lane-per-record read would touch 64 cache lines per instruction), keeping
//                   only the chunks the record's opcode reads; then each lane evaluates its own
//                   record: Quadbox and triangle ops completely; for a distance beat the tree
//                   sum(s) of the beat, which do not depend on the accumulator.  The accumulator
//                   is a strictly sequential float sum over the beats of one vector pair IN
//                   STREAM ORDER (reference
//                   src/main/scala/raytracer_datapath/UnifiedDatapath.scala:1133-1154,
//                   :1028-1064): beats behind the first reset of their opcode inside the wave
//                   start from +0 whatever came before, so the wave finishes them itself.  Left
//                   over are the HEAD beats of the wave (those up to and including its first
//                   reset; all of them when it holds no reset): they continue a pair from an
//                   earlier wave.  The 21-word output records go back through LDS to coalesced
//                   stores.  Per wave and distance opcode the side arrays receive the head
//                   lanes as a 64-bit mask, the head beats' tree sums in stream order (a row
//                   of 64 per wave, filled from its start), the TAIL sum (the running sum
//                   behind the wave's last record, counted from its last reset), and one byte
//                   saying which distance opcodes occur with / without a reset.
//   k_stream_bits   turns the bytes into one bit per wave (64 waves per word), the "opcode
//                   occurs" flags and the index of the last wave that holds the opcode.
//   k_stream_fold   per distance opcode, one fold wave per 64 evaluation waves, one LANE per
//                   evaluation wave.  The sum that runs into a wave is known without any walk
//                   when the nearest wave in front of it that holds the opcode also holds a
//                   reset: it is that wave's tail (the registers entering the stream, in front
//                   of the first one).  Every such wave's head beats are added up by its lane
//                   and patched into the output records: two memory round trips in the common
//                   case, whatever the stream's length.  A wave WITHOUT a reset hands its sum on:
//                   the waves behind it are then walked in order by the whole fold wave (rows
//                   fetched eight waves at a time, the walk itself in registers) up to and
//                   including the next wave that holds a reset; waves without the opcode are
//                   skipped through the bit words, 256 Ki records per step.  Whoever handles the
//                   last wave that holds the opcode writes the registers behind the stream.
// Output fields of foreign opcodes are zero and the two *_reset_accum fields echo the input
// bit for every opcode (:1205, :1254, :1267-1268).
#include "ops.hpp"

#pragma clang fp contract(off)

namespace rtdp {

constexpr int kRecIn = 84;    // words per rt_dp_input_bundle
constexpr int kRecOut = 21;   // words per rt_dp_output_bundle
constexpr int kWaveRecs = 64;

struct StreamSide {
    float *hs[3];                  // [64 * ceil(n / 64)] each: per wave a row of 64 with the tree sums of its head beats in stream order from
                                   //       the row's start: Euclidean sums, angular dots, angular norms
    u4v *heads;                    // [ceil(n / 64)] per wave: head lanes of opcode 2 (x, y = low, high half), of opcode 3 (z, w)
    u4v *tails;                    // [ceil(n / 64)] per wave: tail sums {Euclidean, angular dot, angular norm, unused}; valid where the wave holds a reset
    uint8_t *wavebits;             // [ceil(n / 64)] per wave: 1 / 2 = a beat of opcode 2 / 3 occurs, 4 / 8 = one with its reset bit
    unsigned long long *occ[2];    // [ceil(n / 4096)] bit b of word w: wave 64 w + b holds a beat of opcode 2 / 3
    unsigned long long *rst[2];    // likewise: ... holds one with its reset bit
    uint32_t *flags;               // per launch: [1] opcode 2 present, [2] opcode 3 present, [4] / [5] 1 + the last wave that holds a beat of
                                   // opcode 2 / 3; [8] an opcode > 3 was seen (sticky)
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

// One lane evaluates the record it owns out of its compact row (rc = 11 slots): Quadbox and triangle ops completely, for a
// distance beat the tree sum(s) of the beat, which do not depend on the accumulator.  w: the 21 output words (accumulator
// fields still 0).  Returns false when the opcode is not one of the four.
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

// The wave's output records (live * 21 words, assembled in LDS) to HBM.  A whole wave's 5376 bytes go out in 16-byte pieces when
// the output array is 16-byte aligned (the records of a wave start 5376 = 16 * 336 bytes apart), in coalesced 4-byte pieces when
// it is not: an output array that is a slice of a larger one starts at a multiple of 84 bytes.
__device__ __forceinline__ void stream_store_records(const uint32_t *s_rec, uint32_t *dst, uint32_t live, int lane) {
    if (live == (uint32_t)kWaveRecs && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {  // 1344 words = 336 chunks of 16 bytes
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int g = j * 64 + lane;
            if (g < kWaveRecs * kRecOut / 4) __builtin_nontemporal_store(reinterpret_cast<const u4v *>(s_rec)[g], reinterpret_cast<u4v *>(dst) + g);
        }
    } else if (live == (uint32_t)kWaveRecs) {
#pragma unroll
        for (int k = 0; k < kRecOut; ++k) __builtin_nontemporal_store(s_rec[k * 64 + lane], dst + k * 64 + lane);
    } else {
        for (uint32_t g = lane; g < live * kRecOut; g += 64) dst[g] = s_rec[g];
    }
}

constexpr int kStreamEvalWavesPerCu = 8;  // resident waves of k_stream_eval per CU (environment RT_DP_STREAM_WAVES: another number, 0 = whatever fits)
constexpr int kStageWords = (kWaveRecs * kLiveStride + kWaveRecs) * 4;  // the compact rows and, behind them, one slot per lane for chunks nobody reads

// The wave's records from HBM into the compact rows of s_rec (11 slots of 16 bytes per record); s_op[r] receives record r's
// opcode << 24 | the mask of its live chunks (0 for a record behind the stream's end).  Straight-line code: a chunk that no
// opcode body reads goes to the lane's own spare slot instead of being skipped, so the 21 LDS writes need no branches.
__device__ __forceinline__ void stream_stage(const uint32_t *__restrict__ in, uint64_t r0, uint32_t live, int lane, uint32_t *s_rec, uint32_t *s_op) {
    // Coalesced loads of the wave's whole records (chunk g = j * 64 + lane of its 21 KiB); fetching only the chunks a record's
    // opcode reads measured 9-11 % slower on MI355X -- the lines come in full anyway.  Chunk g belongs to record g / 21; 64 = 3 * 21 + 1,
    // so from one load to the next a lane moves three records on and one chunk up (four records on where the chunk wraps).
    const u4v *src = reinterpret_cast<const u4v *>(in + r0 * kRecIn);
    u4v v[kRecIn / 4];
    if (live == (uint32_t)kWaveRecs) {
#pragma unroll
        for (int j = 0; j < kRecIn / 4; ++j) v[j] = __builtin_nontemporal_load(src + j * 64 + lane);
    } else {
        const uint32_t chunks = live * (kRecIn / 4);
#pragma unroll
        for (int j = 0; j < kRecIn / 4; ++j) {
            const uint32_t g = (uint32_t)j * 64u + lane;
            v[j] = (g < chunks) ? __builtin_nontemporal_load(src + g) : u4v{0u, 0u, 0u, 0u};
        }
    }
    const uint32_t rec0 = (uint32_t)lane / 21u, chunk0 = (uint32_t)lane % 21u;
    {   // Exactly one of a lane's 21 chunks is the first chunk of a record (64 lanes, 64 records): that lane posts the record's opcode.
        const uint32_t jstar = (21u - chunk0) % 21u;
        uint32_t op = 0u;
#pragma unroll
        for (int j = 0; j < kRecIn / 4; ++j) op = (jstar == (uint32_t)j) ? v[j].x : op;
        const uint32_t rstar = rec0 + 3u * jstar + (chunk0 != 0u ? 1u : 0u);
        s_op[rstar] = (rstar < live) ? (((op > 3u ? 0xFFu : op) << 24) | live_chunks(op)) : 0u;  // 0xFF: not an opcode (the 64 lanes cover the 64 records)
        if (lane == 0) s_op[kWaveRecs] = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kRecIn / 4; ++j) {
        const uint32_t wrap = (chunk0 + (uint32_t)j >= 21u) ? 1u : 0u;
        const uint32_t c = chunk0 + (uint32_t)j - 21u * wrap, r = rec0 + 3u * (uint32_t)j + wrap;
        const uint32_t m = s_op[r];  // r <= 64, and entry 64 is 0
        const bool keep = (m >> c) & 1u;
        const uint32_t slot = keep ? (r * kLiveStride + __popc(m & ((1u << c) - 1u))) : ((uint32_t)(kWaveRecs * kLiveStride) + lane);
        reinterpret_cast<u4v *>(s_rec)[slot] = v[j];
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_stream_eval(const uint32_t *__restrict__ in, uint64_t n, int extended, uint32_t *__restrict__ out,
                                                    StreamSide side) {
    __shared__ __align__(16) uint32_t s_rec[kStageWords];  // 12 KiB: live chunks and spare slots; re-used for the output records (5.25 KiB)
    __shared__ uint32_t s_op[kWaveRecs + 1];  // one entry more: the chunk walk of stream_stage ends one record behind the wave's last
    const int lane = threadIdx.x;
    // Workgroups go to the eight XCDs in turn (workgroup i to XCD i mod 8): XCD x works through the x-th eighth of the stream,
    // so that the waves resident on one XCD read neighbouring spans.
    const uint64_t nblk = (n + kWaveRecs - 1) / kWaveRecs, per = (nblk + 7) / 8;
    const uint64_t span = (uint64_t)(blockIdx.x & 7u) * per + (blockIdx.x >> 3);  // = the "wave" index of the side arrays
    if (span >= nblk) return;
    const uint64_t r0 = span * kWaveRecs;
    const uint32_t live = (n - r0 < (uint64_t)kWaveRecs) ? (uint32_t)(n - r0) : (uint32_t)kWaveRecs;

    stream_stage(in, r0, live, lane, s_rec, s_op);

    const bool have = (uint32_t)lane < live;
    uint32_t w[kRecOut], op, reset;
    float sum0, sum1;
    if (!stream_eval_compact(reinterpret_cast<const u4v *>(s_rec) + lane * kLiveStride, s_op[lane] >> 24, have, extended, w, op, reset, sum0, sum1))
        atomicOr(&side.flags[8], 1u);

    // Beats behind the first reset of their opcode in this wave start from +0 whatever came before: their accumulator
    // outputs are final here and travel with the record's coalesced store.  Only the head beats (up to and including the
    // first reset; they continue a pair from an earlier wave) are left to k_stream_fold.
    if (extended) {
        const unsigned long long e2 = __ballot(op == 2u), r2 = __ballot(op == 2u && reset != 0u);
        const unsigned long long e3 = __ballot(op == 3u), r3 = __ballot(op == 3u && reset != 0u);
        unsigned long long h2 = e2, h3 = e3;
        float tail_e = 0.0f, tail_d = 0.0f, tail_n = 0.0f;
        if (r2) {
            const int lo = __ffsll((long long)r2) - 1;
            h2 = e2 & ((lo == 63) ? ~0ull : ((2ull << lo) - 1ull));
            const unsigned long long rest = e2 & ~h2;
            float none0 = 0.0f, res = 0.0f, none1 = 0.0f;
            stream_wave_walk<false>(rest, r2, sum0, 0.0f, lane, tail_e, none0, res, none1);
            if ((rest >> lane) & 1ull) w[16] = __float_as_uint(canon_nan(res));
        }
        if (r3) {
            const int lo = __ffsll((long long)r3) - 1;
            h3 = e3 & ((lo == 63) ? ~0ull : ((2ull << lo) - 1ull));
            const unsigned long long rest = e3 & ~h3;
            float res0 = 0.0f, res1 = 0.0f;
            stream_wave_walk<true>(rest, r3, sum0, sum1, lane, tail_d, tail_n, res0, res1);
            if ((rest >> lane) & 1ull) { w[18] = __float_as_uint(canon_nan(res0)); w[19] = __float_as_uint(canon_nan(res1)); }
        }
        const unsigned long long lower = (1ull << lane) - 1ull;
        if ((h2 >> lane) & 1ull) side.hs[0][r0 + __popcll(h2 & lower)] = sum0;
        if ((h3 >> lane) & 1ull) {
            const int k = __popcll(h3 & lower);
            side.hs[1][r0 + k] = sum0;
            side.hs[2][r0 + k] = sum1;
        }
        // What the fold needs to know about this wave: plain stores, no atomic and no device-scope read in the life of a
        // wave (an earlier form set summary bits and launch flags with atomics guarded by a device-scope load of the flag:
        // one more memory round trip per wave).
        if (lane == 0) {
            side.heads[span] = u4v{(uint32_t)h2, (uint32_t)(h2 >> 32), (uint32_t)h3, (uint32_t)(h3 >> 32)};
            side.tails[span] = u4v{__float_as_uint(tail_e), __float_as_uint(tail_d), __float_as_uint(tail_n), 0u};
            side.wavebits[span] = (uint8_t)((e2 ? 1u : 0u) | (e3 ? 2u : 0u) | (r2 ? 4u : 0u) | (r3 ? 8u : 0u));
        }
    }

    __syncthreads();  // every lane has read its record: the buffer now takes the output records
#pragma unroll
    for (int k = 0; k < kRecOut; ++k) s_rec[lane * kRecOut + k] = w[k];  // stride 21 words is odd: conflict-free
    __syncthreads();
    stream_store_records(s_rec, out + r0 * kRecOut, live, lane);
}

// The per-wave bytes of k_stream_eval become one bit per wave (64 waves per word) and the "opcode occurs" flags.
__global__ __launch_bounds__(256) void k_stream_bits(StreamSide side, uint64_t nblk) {
    const uint64_t w = (uint64_t)blockIdx.x * 256u + threadIdx.x, nwords = (nblk + 63) >> 6;
    unsigned long long m[4] = {0ull, 0ull, 0ull, 0ull};
    if (w < nwords) {
        const u4v *src = reinterpret_cast<const u4v *>(side.wavebits) + 4 * w;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u4v v = src[q];
            const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int wave = 16 * q + b;
                const uint32_t byte = (w * 64 + wave < nblk) ? ((words[b >> 2] >> (8 * (b & 3))) & 0xFFu) : 0u;  // bytes behind the last wave are stale
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] |= (unsigned long long)((byte >> k) & 1u) << wave;
            }
        }
        side.occ[0][w] = m[0];
        side.occ[1][w] = m[1];
        side.rst[0][w] = m[2];
        side.rst[1][w] = m[3];
    }
    __shared__ uint32_t s_last[2];
    if (threadIdx.x < 2) s_last[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (m[k]) atomicMax(&s_last[k], (uint32_t)(w * 64 + (uint64_t)(63 - __builtin_clzll(m[k])) + 1u));
    __syncthreads();
    if (threadIdx.x < 2 && s_last[threadIdx.x]) {
        atomicOr(&side.flags[1 + threadIdx.x], 1u);
        atomicMax(&side.flags[4 + threadIdx.x], s_last[threadIdx.x]);
    }
}

// A value every lane holds alike, told to the compiler (scalar registers, uniform branches).
__device__ __forceinline__ unsigned long long wave_uniform(unsigned long long x) {
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) |
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32);
}

constexpr int kFoldBatch = 8;  // rows in flight per step of a chain walk

__device__ __forceinline__ float pick4(const f4v &q, int i) { return i == 0 ? q.x : i == 1 ? q.y : i == 2 ? q.z : q.w; }

// A pair that runs through whole waves: entered with the running sum c0 / c1 behind a wave without a reset, the waves from j on
// are walked one after the other by the whole wave (the sum is wave-uniform) up to and including the first one that holds a
// reset, or the last one that holds a beat of the opcode at all, which is where the registers behind the stream come from.
template <bool EUC>
__device__ __forceinline__ void stream_fold_chain(const StreamSideView &v, uint64_t nwords, uint64_t j, float c0, float c1, uint64_t last,
                                                  float *__restrict__ state_out, uint32_t *__restrict__ out) {
    constexpr int X = EUC ? 0 : 1;
    const int lane = threadIdx.x;
    const unsigned long long *__restrict__ occ = v.occ[X], *__restrict__ rst = v.rst[X];
    const u4v *__restrict__ heads_of = reinterpret_cast<const u4v *>(v.heads), *__restrict__ tails_of = reinterpret_cast<const u4v *>(v.tails);
    const float *__restrict__ hs0 = v.hs[EUC ? 0 : 1], *__restrict__ hs1 = v.hs[2];
    while (j <= last) {  // every trip moves j on: terminates
        const uint64_t blk = j >> 6, base = blk << 6;
        unsigned long long todo = wave_uniform(occ[blk]) & (~0ull << (j - base));
        if (!todo) {  // nothing of this opcode in the rest of these 64 waves: look at the next 64 words at once
            const uint64_t wl = blk + 1 + lane;
            const unsigned long long far = __ballot(wl < nwords && occ[wl] != 0ull);
            j = far ? ((blk + 1 + (uint64_t)(__ffsll((long long)far) - 1)) << 6) : ((blk + 65) << 6);
            continue;
        }
        const unsigned long long rs = wave_uniform(rst[blk]);
        bool ends = false;
        if (rs & todo) {
            const int sw = __ffsll((long long)(rs & todo)) - 1;
            todo &= (sw == 63) ? ~0ull : ((2ull << sw) - 1ull);
            ends = true;
        }
        const int top = 63 - __builtin_clzll(todo);
        u4v hd = u4v{0u, 0u, 0u, 0u}, tl = u4v{0u, 0u, 0u, 0u};  // lane l: the head lanes and the tails of wave base + l
        if ((todo >> lane) & 1ull) {
            hd = heads_of[base + lane];
            tl = tails_of[base + lane];
        }
        const uint32_t my_lo = EUC ? hd.x : hd.z, my_hi = EUC ? hd.y : hd.w;
        const uint32_t my_t0 = EUC ? tl.x : tl.y, my_t1 = tl.z;
        while (todo) {
            int wv[kFoldBatch];
            float row0[kFoldBatch], row1[kFoldBatch];
#pragma unroll
            for (int k = 0; k < kFoldBatch; ++k) {  // the rows of the next waves that hold beats of the opcode (whole rows exist, their ends are stale)
                wv[k] = todo ? (__ffsll((long long)todo) - 1) : -1;
                todo &= todo - 1;
                const uint64_t idx = (base + (uint64_t)(wv[k] < 0 ? 0 : wv[k])) * 64 + lane;
                row0[k] = (wv[k] >= 0) ? hs0[idx] : 0.0f;
                row1[k] = (!EUC && wv[k] >= 0) ? hs1[idx] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < kFoldBatch; ++k) {
                if (wv[k] < 0) continue;  // wave-uniform
                unsigned long long heads = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)my_lo, wv[k]) |
                                           ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)my_hi, wv[k]) << 32);
                const unsigned long long mask = heads;
                float res0 = 0.0f, res1 = 0.0f;
                for (int r = 0; heads; ++r) {  // the r-th head beat sits in lane L of its wave; its sum is the r-th of the row
                    const int L = __builtin_amdgcn_readfirstlane(__ffsll((long long)heads) - 1);
                    heads &= heads - 1;
                    c0 = c0 + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row0[k]), r));
                    if (lane == L) res0 = c0;
                    if (!EUC) {
                        c1 = c1 + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row1[k]), r));
                        if (lane == L) res1 = c1;
                    }
                }
                if ((mask >> lane) & 1ull) {  // partial-line writes: without nt each one pulls its line into L2 first
                    uint32_t *rec = out + ((base + (uint64_t)wv[k]) * 64 + lane) * kRecOut;
                    if (EUC) {
                        __builtin_nontemporal_store(__float_as_uint(canon_nan(res0)), rec + 16);
                    } else {
                        __builtin_nontemporal_store(__float_as_uint(canon_nan(res0)), rec + 18);
                        __builtin_nontemporal_store(__float_as_uint(canon_nan(res1)), rec + 19);
                    }
                }
            }
        }
        if (ends) {  // behind a reset-holding wave runs what it counted from its last reset on
            c0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)my_t0, top));
            if (!EUC) c1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)my_t1, top));
        }
        if (base + (uint64_t)top == last) {
            if (lane == 0) {
                state_out[0] = canon_nan(c0);
                if (!EUC) state_out[1] = canon_nan(c1);
            }
            return;
        }
        if (ends) return;
        j = base + 64;
    }
}

// The fold of one distance opcode; executed by one wave per 64 evaluation waves (see the head of this file).
template <bool EUC>
__device__ __forceinline__ void stream_fold(const StreamSideView &v, uint64_t n, const float *__restrict__ state_in, float *__restrict__ state_out,
                                            uint32_t *__restrict__ out) {
    constexpr int X = EUC ? 0 : 1;
    const int lane = threadIdx.x;
    const uint64_t nwaves = (n + 63) >> 6, nwords = (nwaves + 63) >> 6;
    const uint64_t g = blockIdx.x, base = g << 6;
    const unsigned long long *__restrict__ occ = v.occ[X], *__restrict__ rst = v.rst[X];
    const u4v *__restrict__ heads_of = reinterpret_cast<const u4v *>(v.heads), *__restrict__ tails_of = reinterpret_cast<const u4v *>(v.tails);
    const float *__restrict__ hs0 = v.hs[EUC ? 0 : 1], *__restrict__ hs1 = v.hs[2];

    const unsigned long long occw = wave_uniform(occ[g]), rstw = wave_uniform(rst[g]);
    const uint64_t last = (uint64_t)v.flags[4 + X] - 1u;  // the caller has seen that the opcode occurs
    // The wave that holds a beat of the opcode nearest in front of this range, and whether it holds a reset; lane l looks at word g - 1 - l.
    long long pw = -1;
    bool pr = true;  // in front of the stream: the registers that enter it count as the tail of a reset-holding wave
    {
        uint64_t w = g;
        while (w > 0) {  // 64 words per trip
            const long long wi = (long long)w - 1 - lane;
            const unsigned long long o = (wi >= 0) ? occ[wi] : 0ull, r = (wi >= 0) ? rst[wi] : 0ull;
            const unsigned long long any = __ballot(o != 0ull);
            if (!any) { w = (w > 64) ? w - 64 : 0; continue; }
            const int L = __ffsll((long long)any) - 1;
            const unsigned long long ow = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)o, L) |
                                          ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o >> 32), L) << 32);
            const unsigned long long rw = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)r, L) |
                                          ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(r >> 32), L) << 32);
            const int bit = 63 - __builtin_clzll(ow);
            pw = (long long)((w - 1 - (uint64_t)L) * 64 + (uint64_t)bit);
            pr = (rw >> bit) & 1ull;
            break;
        }
    }
    if (!occw) return;  // no beat of the opcode in this range: pairs that pass through it are walked by whoever started them

    // Lane l stands for wave base + l.  The sum that runs into a wave is known without walking when the nearest wave in front of
    // it that holds a beat of the opcode also holds a reset: it is that wave's tail.  One lane per such wave walks its head beats.
    const bool mine = (occw >> lane) & 1ull;
    const uint64_t row = (base + lane) * 64;
    u4v hd = u4v{0u, 0u, 0u, 0u}, tl = u4v{0u, 0u, 0u, 0u};
    f4v q0 = f4v{0.0f, 0.0f, 0.0f, 0.0f}, q1 = q0;
    if (mine) {
        hd = heads_of[base + lane];
        tl = tails_of[base + lane];
        q0 = *reinterpret_cast<const f4v *>(hs0 + row);  // the first four head beats' sums; most waves hold no more
        if (!EUC) q1 = *reinterpret_cast<const f4v *>(hs1 + row);
    }
    float f0 = 0.0f, f1 = 0.0f;  // what runs into the first wave of the range that holds a beat, if that is known
    if (pw < 0) {
        f0 = state_in[0];
        if (!EUC) f1 = state_in[1];
    } else if (pr) {
        const u4v t = tails_of[pw];
        f0 = __uint_as_float(EUC ? t.x : t.y);
        f1 = __uint_as_float(t.z);
    }
    const uint32_t my_t0 = EUC ? tl.x : tl.y, my_t1 = tl.z;
    const unsigned long long below = occw & ((1ull << lane) - 1ull);
    const int prev = below ? (63 - __builtin_clzll(below)) : 0;
    const float p0 = __uint_as_float((uint32_t)__shfl((int)my_t0, prev)), p1 = __uint_as_float((uint32_t)__shfl((int)my_t1, prev));
    const bool known = mine && (below ? (((rstw >> prev) & 1ull) != 0) : pr);
    const bool resets = (rstw >> lane) & 1ull;
    float acc0 = below ? p0 : f0, acc1 = below ? p1 : f1;
    unsigned long long m = known ? ((unsigned long long)(EUC ? hd.x : hd.z) | ((unsigned long long)(EUC ? hd.y : hd.w) << 32)) : 0ull;
    for (int r = 0; m; ++r) {
        if (r && !(r & 3)) {  // the next four
            q0 = *reinterpret_cast<const f4v *>(hs0 + row + r);
            if (!EUC) q1 = *reinterpret_cast<const f4v *>(hs1 + row + r);
        }
        const int bt = __ffsll((long long)m) - 1;
        m &= m - 1;
        uint32_t *rec = out + (row + bt) * kRecOut;
        acc0 = acc0 + pick4(q0, r & 3);
        if (EUC) {
            __builtin_nontemporal_store(__float_as_uint(canon_nan(acc0)), rec + 16);  // partial-line writes: without nt each one pulls its line into L2 first
        } else {
            acc1 = acc1 + pick4(q1, r & 3);
            __builtin_nontemporal_store(__float_as_uint(canon_nan(acc0)), rec + 18);
            __builtin_nontemporal_store(__float_as_uint(canon_nan(acc1)), rec + 19);
        }
    }
    if (known && base + lane == last) {  // the registers behind the stream
        state_out[0] = canon_nan(resets ? __uint_as_float(my_t0) : acc0);
        if (!EUC) state_out[1] = canon_nan(resets ? __uint_as_float(my_t1) : acc1);
    }
    // Waves without a reset hand their sum on to the waves behind them: those are walked in order, starting here.
    unsigned long long chains = __ballot(known && !resets && base + lane != last);
    while (chains) {
        const int p = __ffsll((long long)chains) - 1;
        chains &= chains - 1;
        const float c0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc0), p));
        const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc1), p));
        stream_fold_chain<EUC>(v, nwords, base + (uint64_t)p + 1, c0, c1, last, state_out, out);
    }
}

// blockIdx.y = 0: Euclidean, 1: angular, side by side in one launch; an opcode that does not occur in
// the stream (flags written by k_stream_bits) costs one load per workgroup.  state: {euclid, dot, norm}; state_out is
// written completely: by the workgroup that owns the last record of an opcode that occurs, and as a copy of state_in
// for an opcode that does not.
__global__ __launch_bounds__(64) void k_stream_fold(StreamSideView v, uint64_t n, const float *__restrict__ state_in, float *__restrict__ state_out,
                                                    uint32_t *__restrict__ out) {
    if (blockIdx.y == 0) {
        if (v.flags[1]) stream_fold<true>(v, n, state_in, state_out, out);
        else if (blockIdx.x == 0 && threadIdx.x == 0) state_out[0] = state_in[0];
    } else {
        if (v.flags[2]) stream_fold<false>(v, n, state_in + 1, state_out + 1, out);
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
    __shared__ __align__(16) uint32_t s_rec[kStageWords];  // 12 KiB: live chunks and spare slots; re-used for the output records
    __shared__ uint32_t s_op[kWaveRecs + 1];  // one entry more: the chunk walk of stream_stage ends one record behind the wave's last
    const int lane = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * kWaveRecs;
    const uint32_t live = (n - r0 < (uint64_t)kWaveRecs) ? (uint32_t)(n - r0) : (uint32_t)kWaveRecs;
    stream_stage(in, r0, live, lane, s_rec, s_op);
    const bool have = (uint32_t)lane < live;
    uint32_t w[kRecOut], op, reset;
    float sum0, sum1;
    const bool known = stream_eval_compact(reinterpret_cast<const u4v *>(s_rec) + lane * kLiveStride, s_op[lane] >> 24, have, extended, w, op, reset, sum0, sum1);
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
    stream_store_records(s_rec, out + r0 * kRecOut, live, lane);
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

// Workspace for pieces of at most `capacity` records: the two sum arrays, per wave heads / tails / byte, the four bit arrays, 256 bytes of flags.
static StreamSide stream_fused_layout(char *workspace, uint64_t capacity) {
    const uint64_t nblk = (capacity + 63) / 64, nwords = (nblk + 63) / 64;
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    StreamSide side;
    size_t o = 0;
    for (int k = 0; k < 3; ++k) { side.hs[k] = ws_at<float>(workspace, o); o += up(nblk * 64 * 4); }
    side.heads = ws_at<u4v>(workspace, o); o += up(nblk * 16);
    side.tails = ws_at<u4v>(workspace, o); o += up(nblk * 16);
    side.wavebits = ws_at<uint8_t>(workspace, o); o += up(nwords * 64);  // whole 64-byte groups: k_stream_bits reads them with 16-byte loads
    for (int k = 0; k < 2; ++k) { side.occ[k] = ws_at<unsigned long long>(workspace, o); o += up(nwords * 8); }
    for (int k = 0; k < 2; ++k) { side.rst[k] = ws_at<unsigned long long>(workspace, o); o += up(nwords * 8); }
    side.flags = ws_at<uint32_t>(workspace, o);
    return side;
}

size_t stream_fused_workspace(uint64_t capacity) {
    const StreamSide side = stream_fused_layout(nullptr, capacity);
    return static_cast<size_t>(reinterpret_cast<uintptr_t>(side.flags)) + 256;
}

static int stream_eval_waves_asked() {
    static const int waves_env = getenv("RT_DP_STREAM_WAVES") ? atoi(getenv("RT_DP_STREAM_WAVES")) : -1;
    return (waves_env >= 0 && waves_env <= 32) ? waves_env : kStreamEvalWavesPerCu;
}

int stream_eval_waves_per_cu() {
    const int waves = stream_eval_waves_asked();
    int achieved = 0;
    if (waves > 0) (void)residency_pad(reinterpret_cast<const void *>(&k_stream_eval), 64, waves, &achieved);
    return achieved;
}

// One piece of n <= capacity records.  first_piece: also clears the sticky "opcode > 3" word, once per call.
hipError_t stream_fused_eval(const uint32_t *in, uint64_t n, uint64_t capacity, int extended, uint32_t *out, char *workspace,
                             StreamSideView *view, hipStream_t s, bool first_piece) {
    const StreamSide side = stream_fused_layout(workspace, capacity);
    const uint64_t nblk = (n + 63) / 64;
    if (n > capacity || nblk > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(side.flags, 0, first_piece ? 256 : 32, s);  // the per-piece flags (all flags on the first piece)
    if (e != hipSuccess) return e;
    // Eight or nine waves per CU: with fewer the kernel's instructions are not covered, with more the memory system sees more
    // open 21 KiB spans than it serves well (MI355X, ms per call at 2^22 / 2^24 / 2^26 records: 7 waves 0.377 / 1.27 / 4.85,
    // 8: 0.355 / 1.22 / 4.85, 9: 0.340 / 1.22 / 4.90, 12: 0.350 / 1.30 / 5.36).
    const int waves = stream_eval_waves_asked();
    const unsigned pad = waves > 0 ? residency_pad(reinterpret_cast<const void *>(&k_stream_eval), 64, waves) : 0u;
    hipLaunchKernelGGL(k_stream_eval, dim3((unsigned)(((nblk + 7) / 8) * 8)), dim3(64), pad, s, in, n, extended, out, side);
    if (extended) hipLaunchKernelGGL(k_stream_bits, dim3((unsigned)(((nblk + 63) / 64 + 255) / 256)), dim3(256), 0, s, side, nblk);
    view->hs[0] = side.hs[0]; view->hs[1] = side.hs[1]; view->hs[2] = side.hs[2]; view->heads = side.heads; view->tails = side.tails;
    view->occ[0] = side.occ[0]; view->occ[1] = side.occ[1]; view->rst[0] = side.rst[0]; view->rst[1] = side.rst[1]; view->flags = side.flags;
    return hipGetLastError();
}

hipError_t stream_fused_fold(const StreamSideView &v, uint64_t n, const float *state_in, float *state_out, uint32_t *out, hipStream_t s) {
    const uint64_t blocks = ((n + 63) / 64 + 63) / 64;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_stream_fold, dim3((unsigned)blocks, 2), dim3(64), 0, s, v, n, state_in, state_out, out);
    return hipGetLastError();
}

}  // namespace rtdp
