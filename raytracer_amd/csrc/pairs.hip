// The format step on either side of the distance ops: vector pairs of any length <-> beats.
// Follows the reference's job splitters (src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:213-269
// Euclidean, :271-308 angular): a pair of `len` elements becomes ceil(len / width) beats (width 16 or 8), the
// last one zero-padded, mask = the low min(width, remaining) bits, reset bit on the last beat; an empty pair
// yields no beat.  The way back picks, per pair, the value its last beat produced, which is what the
// reference's tests read (Datapath_test.scala:287-290, :353-356: outputs whose reset bit is set).
//   k_pairs_to_beats   one thread per beat element; the pair of a beat by binary search in the beat offsets
//   k_beats_to_pairs   one thread per pair
#include "kernels.h"

namespace rtdp {

namespace {

// largest p with offsets[p] <= x, offsets ascending with offsets[0] = 0 and x < offsets[n]
__device__ __forceinline__ uint64_t owner(const uint64_t *__restrict__ offsets, uint64_t n, uint64_t x) {
    uint64_t lo = 0, hi = n;  // offsets[lo] <= x < offsets[hi]
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (offsets[mid] <= x) lo = mid;
        else hi = mid;
    }
    return lo;
}

template <int W>
__global__ __launch_bounds__(256) void k_pairs_to_beats(const float *__restrict__ a, const float *__restrict__ b,
                                                       const uint64_t *__restrict__ pair_offsets, const uint64_t *__restrict__ beat_offsets,
                                                       uint64_t n_pairs, uint64_t n_beats, float *__restrict__ out_a, float *__restrict__ out_b,
                                                       uint32_t *__restrict__ out_ctl) {
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t beat = g / W;
    const uint32_t col = (uint32_t)(g % W);
    if (beat >= n_beats) return;
    const uint64_t p = owner(beat_offsets, n_pairs, beat);
    const uint64_t in_pair = beat - beat_offsets[p];
    const uint64_t first = pair_offsets[p], len = pair_offsets[p + 1] - first;
    const uint64_t done = in_pair * W;
    const uint32_t valid = (len - done < (uint64_t)W) ? (uint32_t)(len - done) : (uint32_t)W;
    const bool on = col < valid;
    out_a[g] = on ? a[first + done + col] : 0.0f;
    out_b[g] = on ? b[first + done + col] : 0.0f;
    if (col == 0) out_ctl[beat] = ((1u << valid) - 1u) | ((beat + 1 == beat_offsets[p + 1]) ? (1u << 16) : 0u);
}

__global__ __launch_bounds__(256) void k_beats_to_pairs(const float *__restrict__ values, const uint64_t *__restrict__ beat_offsets,
                                                       uint64_t n_pairs, float *__restrict__ out) {
    const uint64_t p = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (p >= n_pairs) return;
    const uint64_t a = beat_offsets[p], e = beat_offsets[p + 1];
    out[p] = (e > a) ? values[e - 1] : 0.0f;  // an empty pair has no beat: the accumulator's reset value
}

}  // namespace

hipError_t pairs_to_beats(const float *a, const float *b, const uint64_t *pair_offsets, const uint64_t *beat_offsets, uint64_t n_pairs,
                          uint64_t n_beats, uint32_t width, float *out_a, float *out_b, uint32_t *out_ctl, hipStream_t s) {
    if (n_beats == 0) return hipSuccess;
    const uint64_t blocks = (n_beats * width + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (width == 16) hipLaunchKernelGGL(k_pairs_to_beats<16>, dim3((unsigned)blocks), dim3(256), 0, s, a, b, pair_offsets, beat_offsets, n_pairs, n_beats, out_a, out_b, out_ctl);
    else hipLaunchKernelGGL(k_pairs_to_beats<8>, dim3((unsigned)blocks), dim3(256), 0, s, a, b, pair_offsets, beat_offsets, n_pairs, n_beats, out_a, out_b, out_ctl);
    return hipGetLastError();
}

hipError_t beats_to_pairs(const float *values, const uint64_t *beat_offsets, uint64_t n_pairs, float *out, hipStream_t s) {
    if (n_pairs == 0) return hipSuccess;
    const uint64_t blocks = (n_pairs + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_beats_to_pairs, dim3((unsigned)blocks), dim3(256), 0, s, values, beat_offsets, n_pairs, out);
    return hipGetLastError();
}

}  // namespace rtdp
