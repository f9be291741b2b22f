// Stream front-end: record-at-a-time streams (EnhancedInputBundle images in arrival
// order, any opcode mix) that already live in HBM are partitioned by opcode on the
// device, evaluated by the per-opcode kernels, and the results are written back as
// EnhancedOutputBundle images in arrival order.
//
//   k_stream_count    per 256-record block: how many records of each opcode
//   k_stream_scan     exclusive scan of those counts over the blocks (per opcode)
//   k_stream_scatter  stable partition: record i goes to slot rank(i) of its opcode's
//                     structure-of-arrays segment; slot[i] remembers (opcode, reset, rank)
//   (per-opcode kernels of kernels.hip run on the segments)
//   k_stream_gather   output record i is assembled from its opcode's result arrays
//
// The partition is stable, so each accumulator register sees its beats in stream order
// (reference src/main/scala/raytracer_datapath/UnifiedDatapath.scala:1133-1154, :1028-1064).
// Output fields of foreign opcodes are zero and the two *_reset_accum fields echo the
// input bit for every opcode (:1205, :1254, :1267-1268).
#include "kernels.h"

namespace rtdp {

constexpr int kInWords = 84;   // words per rt_dp_input_bundle
constexpr int kOutWords = 21;  // words per rt_dp_output_bundle
constexpr int kBlk = 256;

__global__ __launch_bounds__(kBlk) void k_stream_count(const uint32_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ blockcnt,
                                                        uint32_t *__restrict__ bad_opcode) {
    __shared__ uint32_t s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * kBlk + threadIdx.x;
    uint32_t op = 0xFFFFFFFFu;
    if (i < n) {
        op = in[i * kInWords];
        if (op > 3u) atomicOr(bad_opcode, 1u);  // cannot occur in the RTL's 2-bit field
    }
#pragma unroll
    for (uint32_t o = 0; o < 4; ++o) {
        const unsigned long long m = __ballot(op == o);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cnt[o], (uint32_t)__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x < 4) blockcnt[(uint64_t)blockIdx.x * 4 + threadIdx.x] = s_cnt[threadIdx.x];
}

// One workgroup; counts [nblk][4] become exclusive offsets in place, totals[4] receives the sums.
__global__ __launch_bounds__(1024) void k_stream_scan(uint32_t *__restrict__ blockcnt, uint32_t nblk, uint32_t *__restrict__ totals) {
    __shared__ uint32_t s_part[1024][4];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nblk + 1023u) / 1024u;
    const uint32_t b0 = t * per, b1 = (b0 + per < nblk) ? b0 + per : nblk;
    uint32_t sum[4] = {0, 0, 0, 0};
    for (uint32_t b = b0; b < b1; ++b)
#pragma unroll
        for (int o = 0; o < 4; ++o) sum[o] += blockcnt[(uint64_t)b * 4 + o];
#pragma unroll
    for (int o = 0; o < 4; ++o) s_part[t][o] = sum[o];
    __syncthreads();
    if (t < 4) {  // serial scan of the 1024 partial sums, one opcode per lane
        uint32_t run = 0;
        for (int k = 0; k < 1024; ++k) { const uint32_t v = s_part[k][t]; s_part[k][t] = run; run += v; }
        totals[t] = run;
    }
    __syncthreads();
    uint32_t run[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) run[o] = s_part[t][o];
    for (uint32_t b = b0; b < b1; ++b)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const uint32_t v = blockcnt[(uint64_t)b * 4 + o];
            blockcnt[(uint64_t)b * 4 + o] = run[o];
            run[o] += v;
        }
}

// The box and triangle segments behind the partition are tiled segments (rt_dp.h: [tile][word][kTileOps]): pointer f is the
// run of word f in tile 0, and op r of a segment with W words per op lives at offset tiled_at(r, W) from it.
__device__ __forceinline__ uint64_t tiled_at(uint32_t r, uint32_t words) {
    return (uint64_t)(r / kTileOps) * kTileOps * words + (r % kTileOps);
}

struct StreamSegs {  // writable images of the per-opcode operand arrays
    float *box[kBoxFields];
    float *tri[kTriFields];
    uint32_t *tri_k;
    float *euc_a, *euc_b;
    uint32_t *euc_ctl;
    float *ang_a, *ang_b;
    uint32_t *ang_ctl;
};

__global__ __launch_bounds__(kBlk) void k_stream_scatter(const uint32_t *__restrict__ in, uint64_t n, const uint32_t *__restrict__ blockoff,
                                                          uint32_t *__restrict__ slot, StreamSegs seg) {
    __shared__ uint32_t s_wave[4][4];  // [wave][opcode]: records of that opcode in the wave
    const uint64_t i = (uint64_t)blockIdx.x * kBlk + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t *rec = in + (i < n ? i : 0) * kInWords;
    const uint32_t op = (i < n) ? rec[0] : 0xFFFFFFFFu;
    uint32_t before = 0;  // same-opcode records ahead of this one inside the wave
#pragma unroll
    for (uint32_t o = 0; o < 4; ++o) {
        const unsigned long long m = __ballot(op == o);
        if (op == o) before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave][o] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (i >= n || op > 3u) return;
    uint32_t r = blockoff[(uint64_t)blockIdx.x * 4 + op] + before;
    for (int w = 0; w < wave; ++w) r += s_wave[w][op];
    const uint32_t reset = rec[83] & 1u;
    slot[i] = op | (reset << 2) | (r << 3);

    if (op == 1u) {  // Quadbox: origin, dir, inv, extent, four AABBs
        const uint64_t at = tiled_at(r, kBoxFields);
#pragma unroll
        for (int f = 0; f < 10; ++f) seg.box[f][at] = __uint_as_float(rec[1 + f]);
#pragma unroll
        for (int f = 0; f < 24; ++f) seg.box[10 + f][at] = __uint_as_float(rec[17 + f]);
    } else if (op == 0u) {  // Triangle: origin, shear, vertices, selectors
        const uint64_t at = tiled_at(r, kTriInWords);
#pragma unroll
        for (int f = 0; f < 3; ++f) seg.tri[f][at] = __uint_as_float(rec[1 + f]);
#pragma unroll
        for (int f = 0; f < 3; ++f) seg.tri[3 + f][at] = __uint_as_float(rec[14 + f]);
#pragma unroll
        for (int f = 0; f < 9; ++f) seg.tri[6 + f][at] = __uint_as_float(rec[41 + f]);
        seg.tri_k[at] = (rec[11] & 3u) | ((rec[12] & 3u) << 2) | ((rec[13] & 3u) << 4);
    } else if (op == 2u) {  // Euclidean beat
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            seg.euc_a[(uint64_t)r * 16 + e] = __uint_as_float(rec[50 + e]);
            seg.euc_b[(uint64_t)r * 16 + e] = __uint_as_float(rec[66 + e]);
        }
        seg.euc_ctl[r] = (rec[82] & 0xFFFFu) | (reset << 16);
    } else {  // Angular beat: query = a[0..7], candidate = b[0..7]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            seg.ang_a[(uint64_t)r * 8 + e] = __uint_as_float(rec[50 + e]);
            seg.ang_b[(uint64_t)r * 8 + e] = __uint_as_float(rec[66 + e]);
        }
        seg.ang_ctl[r] = (rec[82] & 0xFFFFu) | (reset << 16);
    }
}

struct StreamResults {
    const float *tmin[4];
    const uint32_t *flags;
    const float *t_num, *t_denom;
    const uint32_t *hit;
    const float *euc;
    const float *dot, *norm;
};

__global__ __launch_bounds__(kBlk) void k_stream_gather(const uint32_t *__restrict__ slot, uint64_t n, StreamResults res, int extended,
                                                         uint32_t *__restrict__ out) {
    __shared__ uint32_t s_rec[kBlk * kOutWords];  // 21 KiB; stride 21 words is odd: conflict-free
    const uint64_t base = (uint64_t)blockIdx.x * kBlk;
    const uint64_t i = base + threadIdx.x;
    uint32_t *w = s_rec + threadIdx.x * kOutWords;
#pragma unroll
    for (int k = 0; k < kOutWords; ++k) w[k] = 0u;
    if (i < n) {
        const uint32_t s = slot[i];
        const uint32_t op = s & 3u, reset = (s >> 2) & 1u, r = s >> 3;
        w[0] = op;
        if (extended) { w[17] = reset; w[20] = reset; }
        if (op == 1u) {
            const uint64_t at = tiled_at(r, kBoxOutWords);
            const uint32_t fl = res.flags[at];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                w[1 + k] = __float_as_uint(res.tmin[k][at]);
                w[5 + k] = (fl >> k) & 1u;
                w[9 + k] = (fl >> (4 + 2 * k)) & 3u;
            }
        } else if (op == 0u) {
            const uint64_t at = tiled_at(r, kTriOutWords);
            w[13] = __float_as_uint(res.t_num[at]);
            w[14] = __float_as_uint(res.t_denom[at]);
            w[15] = res.hit[at];
        } else if (op == 2u) {
            if (extended) w[16] = __float_as_uint(res.euc[r]);
        } else {
            if (extended) { w[18] = __float_as_uint(res.dot[r]); w[19] = __float_as_uint(res.norm[r]); }
        }
    }
    __syncthreads();
    // the block's records are contiguous in the output: coalesced dword stores
    const uint64_t live = (n - base < (uint64_t)kBlk) ? (n - base) : kBlk;
    const uint32_t words = (uint32_t)live * kOutWords;
    for (uint32_t k = threadIdx.x; k < words; k += kBlk) out[base * kOutWords + k] = s_rec[k];
}

hipError_t stream_count_scan(const uint32_t *in, uint64_t n, uint32_t *blockcnt, uint32_t *totals_and_flag, hipStream_t s) {
    const uint32_t nblk = (uint32_t)((n + kBlk - 1) / kBlk);
    hipError_t e = hipMemsetAsync(totals_and_flag, 0, 8 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_stream_count, dim3(nblk), dim3(kBlk), 0, s, in, n, blockcnt, totals_and_flag + 4);
    hipLaunchKernelGGL(k_stream_scan, dim3(1), dim3(1024), 0, s, blockcnt, nblk, totals_and_flag);
    return hipGetLastError();
}

hipError_t stream_scatter(const uint32_t *in, uint64_t n, const uint32_t *blockoff, uint32_t *slot, const BoxArgs &box, const TriArgs &tri,
                          const VecArgs &euc, const VecArgs &ang, hipStream_t s) {
    StreamSegs seg;
    for (int f = 0; f < kBoxFields; ++f) seg.box[f] = const_cast<float *>(box.in[f]);
    for (int f = 0; f < kTriFields; ++f) seg.tri[f] = const_cast<float *>(tri.in[f]);
    seg.tri_k = const_cast<uint32_t *>(tri.k);
    seg.euc_a = const_cast<float *>(euc.a); seg.euc_b = const_cast<float *>(euc.b); seg.euc_ctl = const_cast<uint32_t *>(euc.ctl);
    seg.ang_a = const_cast<float *>(ang.a); seg.ang_b = const_cast<float *>(ang.b); seg.ang_ctl = const_cast<uint32_t *>(ang.ctl);
    const uint32_t nblk = (uint32_t)((n + kBlk - 1) / kBlk);
    hipLaunchKernelGGL(k_stream_scatter, dim3(nblk), dim3(kBlk), 0, s, in, n, blockoff, slot, seg);
    return hipGetLastError();
}

hipError_t stream_gather(const uint32_t *slot, uint64_t n, const BoxArgs &box, const TriArgs &tri, const VecArgs &euc, const VecArgs &ang,
                         int extended, uint32_t *out, hipStream_t s) {
    StreamResults r;
    for (int k = 0; k < 4; ++k) r.tmin[k] = box.tmin[k];
    r.flags = box.flags;
    r.t_num = tri.t_num; r.t_denom = tri.t_denom; r.hit = tri.hit;
    r.euc = euc.out0;
    r.dot = ang.out0; r.norm = ang.out1;
    const uint32_t nblk = (uint32_t)((n + kBlk - 1) / kBlk);
    hipLaunchKernelGGL(k_stream_gather, dim3(nblk), dim3(kBlk), 0, s, slot, n, r, extended, out);
    return hipGetLastError();
}

}  // namespace rtdp
