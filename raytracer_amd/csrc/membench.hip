// HBM streaming calibration for gfx950: what a plain read / copy kernel reaches
// on this box with the access widths the datapath kernels use.  Two uses:
//   * the measured ceiling next to which the kernels' achieved GB/s is read;
//   * known byte counts per kernel to calibrate rocprofv3's FETCH_SIZE /
//     WRITE_SIZE for 4-, 8- and 16-byte-per-lane accesses.
// Build: hipcc --offload-arch=gfx950 -O3 membench.hip -o membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <algorithm>

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static unsigned g_lds = 0;  // dynamic LDS per workgroup, used only to limit residency

template <typename T, bool NT, int STREAMS>
__global__ __launch_bounds__(256) void k_read(const T *const in, T *sink, size_t n_per_stream) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_per_stream) return;
    T acc = T(0);
#pragma unroll
    for (int s = 0; s < STREAMS; ++s) {
        const T *p = in + (size_t)s * n_per_stream + i;
        acc += NT ? __builtin_nontemporal_load(p) : *p;
    }
    // never true for the fill pattern: keeps the loads alive without writing
    if (reinterpret_cast<float *>(&acc)[0] == 12345.678f) sink[i] = acc;
}

template <typename T, bool NT>
__global__ __launch_bounds__(256) void k_copy(const T *in, T *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
    else out[i] = in[i];
}

template <typename T, bool NT, int STREAMS>
__global__ __launch_bounds__(256) void k_write(T *out, size_t n_per_stream) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_per_stream) return;
    T v = T((float)threadIdx.x);
#pragma unroll
    for (int s = 0; s < STREAMS; ++s) {
        T *p = out + (size_t)s * n_per_stream + i;
        if (NT) __builtin_nontemporal_store(v, p);
        else *p = v;
    }
}

// The box kernel's memory pattern without its arithmetic: RS read streams and WS write
// streams of one element of type T per lane.  WIDE: lane pairs exchange their results so
// that every store instruction writes 2 x sizeof(T) per lane on half of the arrays.
template <typename T, int RS, int WS>
__global__ __launch_bounds__(256) void k_mix(const T *in, T *out, size_t n_per_stream) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_per_stream) return;
    T acc = T(0);
#pragma unroll
    for (int s = 0; s < RS; ++s) acc += __builtin_nontemporal_load(in + (size_t)s * n_per_stream + i);
#pragma unroll
    for (int s = 0; s < WS; ++s) __builtin_nontemporal_store(acc, out + (size_t)s * n_per_stream + i);
}

template <int RS, int WS>
__global__ __launch_bounds__(256) void k_mix_wide(const f2v *in, f2v *out, size_t n_per_stream) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_per_stream) return;  // n_per_stream is a multiple of 256 in this benchmark
    f2v acc = f2v(0);
#pragma unroll
    for (int s = 0; s < RS; ++s) acc += __builtin_nontemporal_load(in + (size_t)s * n_per_stream + i);
    // partner's value through DPP quad_perm [1,0,3,2]
    f2v oth;
    oth.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.x), 0xB1, 0xF, 0xF, true));
    oth.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.y), 0xB1, 0xF, 0xF, true));
    const bool odd = threadIdx.x & 1;
    f4v w;
    w.x = odd ? oth.x : acc.x; w.y = odd ? oth.y : acc.y; w.z = odd ? acc.x : oth.x; w.w = odd ? acc.y : oth.y;
    const size_t pair = i & ~(size_t)1;
#pragma unroll
    for (int s = 0; s < WS; s += 2) {
        const int mine = s + (odd ? 1 : 0);
        if (mine < WS) __builtin_nontemporal_store(w, reinterpret_cast<f4v *>(out + (size_t)mine * n_per_stream + pair));
    }
}

template <typename F> static double time_ms(F launch, int reps = 7) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ts;
    launch();
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

template <typename T, bool NT, int STREAMS> static void run_read(const char *name, const void *buf, void *sink, size_t bytes) {
    const size_t n = bytes / sizeof(T) / STREAMS;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    double ms = time_ms([&] { hipLaunchKernelGGL((k_read<T, NT, STREAMS>), dim3(blocks), dim3(256), g_lds, 0, (const T *)buf, (T *)sink, n); });
    printf("%-28s %8.1f GB/s  (%.3f ms, %zu bytes read)\n", name, n * STREAMS * sizeof(T) / ms / 1e6, ms, n * STREAMS * sizeof(T));
}

template <typename T, bool NT> static void run_copy(const char *name, const void *in, void *out, size_t bytes) {
    const size_t n = bytes / sizeof(T);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    double ms = time_ms([&] { hipLaunchKernelGGL((k_copy<T, NT>), dim3(blocks), dim3(256), g_lds, 0, (const T *)in, (T *)out, n); });
    printf("%-28s %8.1f GB/s  (%.3f ms, %zu bytes read + %zu written)\n", name, 2.0 * n * sizeof(T) / ms / 1e6, ms, n * sizeof(T), n * sizeof(T));
}

template <typename T, bool NT, int STREAMS> static void run_write(const char *name, void *buf, size_t bytes) {
    const size_t n = bytes / sizeof(T) / STREAMS;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    double ms = time_ms([&] { hipLaunchKernelGGL((k_write<T, NT, STREAMS>), dim3(blocks), dim3(256), g_lds, 0, (T *)buf, n); });
    printf("%-28s %8.1f GB/s  (%.3f ms, %zu bytes written)\n", name, n * STREAMS * sizeof(T) / ms / 1e6, ms, n * STREAMS * sizeof(T));
}

template <typename T, int RS, int WS, bool WIDE> static void run_mix(const char *name, const void *in, void *out, size_t bytes) {
    size_t n = bytes / sizeof(T) / RS;
    n &= ~(size_t)255;
    const unsigned blocks = (unsigned)(n / 256);
    double ms;
    if constexpr (WIDE) ms = time_ms([&] { hipLaunchKernelGGL((k_mix_wide<RS, WS>), dim3(blocks), dim3(256), g_lds, 0, (const f2v *)in, (f2v *)out, n); });
    else ms = time_ms([&] { hipLaunchKernelGGL((k_mix<T, RS, WS>), dim3(blocks), dim3(256), g_lds, 0, (const T *)in, (T *)out, n); });
    const double b = (double)n * sizeof(T) * (RS + WS);
    printf("%-28s %8.1f GB/s  (%.3f ms, %d read + %d write streams)\n", name, b / ms / 1e6, ms, RS, WS);
}


// The record-stream pattern: a wave reads RECS consecutive 336-byte records (coalesced 16-byte loads, chunk g = j * 64 + lane)
// and writes RECS 84-byte records; THREADS per workgroup, every wave with a span of its own.  No arithmetic beyond what keeps
// the loads alive.
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
// The same with the loads of a wave issued in GROUPS portions, each waited for before the next goes out: fewer bytes in
// flight per wave at the price of GROUPS memory round trips in the life of a wave.
template <int GROUPS>
__global__ __launch_bounds__(64) void k_rec_grouped(const u4v *in, u4v *out, size_t n_rec) {
    const int lane = threadIdx.x;
    const size_t r0 = (size_t)blockIdx.x * 64;
    if (r0 >= n_rec) return;
    const u4v *src = in + r0 * 21;
    u4v acc = u4v{0u, 0u, 0u, 0u};
    constexpr int PER = 21 / GROUPS;
#pragma unroll
    for (int g0 = 0; g0 < GROUPS; ++g0) {
        u4v v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) v[j] = __builtin_nontemporal_load(src + (g0 * PER + j) * 64 + lane);
#pragma unroll
        for (int j = 0; j < PER; ++j) acc ^= v[j];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    u4v *dst = out + (r0 * 21) / 4;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int g = j * 64 + lane;
        if (g < 336) __builtin_nontemporal_store(acc, dst + g);
    }
}

template <int GROUPS> static void run_rec_grouped(const char *name, const void *in, void *out, size_t n_rec, unsigned lds) {
    const unsigned blocks = (unsigned)((n_rec + 63) / 64);
    double ms = time_ms([&] { hipLaunchKernelGGL((k_rec_grouped<GROUPS>), dim3(blocks), dim3(64), lds, 0, (const u4v *)in, (u4v *)out, n_rec); });
    printf("%-44s lds %6u  %8.1f GB/s  (%.3f ms, 420 B per record)\n", name, lds, n_rec * 420.0 / ms / 1e6, ms);
}

template <int THREADS, int RECS, bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(THREADS) void k_rec(const u4v *in, u4v *out, size_t n_rec) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    const size_t r0 = wave * RECS;
    if (r0 >= n_rec) return;
    constexpr int kIn = RECS * 21, kOut = (RECS * 21) / 4;  // chunks of 16 bytes
    const u4v *src = in + r0 * 21;
    u4v acc = u4v{0u, 0u, 0u, 0u};
    u4v v[(kIn + 63) / 64];
#pragma unroll
    for (int j = 0; j < (kIn + 63) / 64; ++j) {
        const int g = j * 64 + lane;
        v[j] = u4v{0u, 0u, 0u, 0u};
        if (g < kIn) v[j] = NT_LD ? __builtin_nontemporal_load(src + g) : src[g];
    }
#pragma unroll
    for (int j = 0; j < (kIn + 63) / 64; ++j) acc ^= v[j];
    u4v *dst = out + (r0 * 21) / 4;
#pragma unroll
    for (int j = 0; j < (kOut + 63) / 64; ++j) {
        const int g = j * 64 + lane;
        if (g < kOut) {
            if (NT_ST) __builtin_nontemporal_store(acc, dst + g);
            else dst[g] = acc;
        }
    }
}

template <int THREADS, int RECS, bool NT_LD, bool NT_ST> static void run_rec(const char *name, const void *in, void *out, size_t n_rec, unsigned lds) {
    const size_t waves = (n_rec + RECS - 1) / RECS;
    const unsigned blocks = (unsigned)((waves + THREADS / 64 - 1) / (THREADS / 64));
    double ms = time_ms([&] { hipLaunchKernelGGL((k_rec<THREADS, RECS, NT_LD, NT_ST>), dim3(blocks), dim3(THREADS), lds, 0, (const u4v *)in, (u4v *)out, n_rec); });
    printf("%-44s lds %6u  %8.1f GB/s  (%.3f ms, 420 B per record)\n", name, lds, n_rec * 420.0 / ms / 1e6, ms);
}

static void rec_probes(void *a, void *b, size_t bytes) {
    size_t n_rec = bytes / 336;
    if (n_rec > ((size_t)1 << 24)) n_rec = (size_t)1 << 24;
    const unsigned caps[] = {0u, 11520u, 21504u, 32768u, 65536u};
    for (unsigned lds : caps) {
        run_rec<64, 64, true, true>("rec 64 thr, 64 rec/wave, nt ld, nt st", a, b, n_rec, lds);
        run_rec<64, 64, false, true>("rec 64 thr, 64 rec/wave, ld, nt st", a, b, n_rec, lds);
        run_rec<64, 64, true, false>("rec 64 thr, 64 rec/wave, nt ld, st", a, b, n_rec, lds);
        run_rec<64, 64, false, false>("rec 64 thr, 64 rec/wave, ld, st", a, b, n_rec, lds);
        run_rec<256, 64, true, true>("rec 256 thr, 64 rec/wave, nt ld, nt st", a, b, n_rec, lds);
        run_rec<64, 32, true, true>("rec 64 thr, 32 rec/wave, nt ld, nt st", a, b, n_rec, lds);
        run_rec<256, 32, true, true>("rec 256 thr, 32 rec/wave, nt ld, nt st", a, b, n_rec, lds);
        run_rec<64, 16, true, true>("rec 64 thr, 16 rec/wave, nt ld, nt st", a, b, n_rec, lds);
        run_rec<256, 16, true, true>("rec 256 thr, 16 rec/wave, nt ld, nt st", a, b, n_rec, lds);
        run_rec_grouped<1>("rec 64 thr, 64 rec/wave, 1 group", a, b, n_rec, lds);
        run_rec_grouped<3>("rec 64 thr, 64 rec/wave, 3 groups", a, b, n_rec, lds);
        run_rec_grouped<7>("rec 64 thr, 64 rec/wave, 7 groups", a, b, n_rec, lds);
    }
}

// The record-stream pattern with arithmetic behind the loads (WORK dependent multiply-adds per loaded chunk) in workgroups of
// WAVES waves; with TOKEN at most one wave of a workgroup has loads in flight at a time (a lock in LDS): do fewer, shorter load
// phases per CU (a smaller window of open DRAM pages) pay when many waves are needed to cover the arithmetic?
template <int WAVES, bool TOKEN, int WORK>
__global__ __launch_bounds__(WAVES * 64) void k_rec_token(const u4v *in, u4v *out, size_t n_rec) {
    __shared__ int s_lock;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_lock = 0;
    __syncthreads();
    const size_t wave = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const size_t r0 = wave * 64;
    if (r0 >= n_rec) return;
    const u4v *src = in + r0 * 21;
    u4v v[21];
    if (TOKEN) {
        if (lane == 0) while (atomicCAS(&s_lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int j = 0; j < 21; ++j) v[j] = __builtin_nontemporal_load(src + j * 64 + lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (TOKEN) {
        if (lane == 0) atomicExch(&s_lock, 0);
    }
    u4v acc = u4v{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 21; ++j) {
        u4v t = v[j];
        for (int k = 0; k < WORK; ++k) t = t * 1664525u + 1013904223u;
        acc ^= t;
    }
    u4v *dst = out + (r0 * 21) / 4;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int g = j * 64 + lane;
        if (g < 336) __builtin_nontemporal_store(acc, dst + g);
    }
}

template <int WAVES, bool TOKEN, int WORK> static void run_rec_token(const char *name, const void *in, void *out, size_t n_rec, unsigned lds) {
    const unsigned blocks = (unsigned)((n_rec / 64 + WAVES - 1) / WAVES);
    double ms = time_ms([&] { hipLaunchKernelGGL((k_rec_token<WAVES, TOKEN, WORK>), dim3(blocks), dim3(WAVES * 64), lds, 0, (const u4v *)in, (u4v *)out, n_rec); });
    printf("%-44s lds %6u  %8.1f GB/s  (%.3f ms, 420 B per record)\n", name, lds, n_rec * 420.0 / ms / 1e6, ms);
}

static void token_probes(void *a, void *b, size_t bytes) {
    size_t n_rec = bytes / 336;
    if (n_rec > ((size_t)1 << 24)) n_rec = (size_t)1 << 24;
    // one workgroup per CU (12 and 8 waves), two (6 waves each), three (4 each), residency forced by unused LDS
    run_rec_token<6, false, 0>("6 waves x2, free, no work", a, b, n_rec, 66560);
    run_rec_token<6, true, 0>("6 waves x2, token, no work", a, b, n_rec, 66560);
    run_rec_token<6, false, 8>("6 waves x2, free, work 8", a, b, n_rec, 66560);
    run_rec_token<6, true, 8>("6 waves x2, token, work 8", a, b, n_rec, 66560);
    run_rec_token<6, false, 16>("6 waves x2, free, work 16", a, b, n_rec, 66560);
    run_rec_token<6, true, 16>("6 waves x2, token, work 16", a, b, n_rec, 66560);
    run_rec_token<4, false, 8>("4 waves x3, free, work 8", a, b, n_rec, 43520);
    run_rec_token<4, true, 8>("4 waves x3, token, work 8", a, b, n_rec, 43520);
    run_rec_token<4, false, 16>("4 waves x3, free, work 16", a, b, n_rec, 43520);
    run_rec_token<4, true, 16>("4 waves x3, token, work 16", a, b, n_rec, 43520);
    run_rec_token<12, false, 8>("12 waves x1, free, work 8", a, b, n_rec, 133120);
    run_rec_token<12, true, 8>("12 waves x1, token, work 8", a, b, n_rec, 133120);
    run_rec_token<12, false, 16>("12 waves x1, free, work 16", a, b, n_rec, 133120);
    run_rec_token<12, true, 16>("12 waves x1, token, work 16", a, b, n_rec, 133120);
    run_rec_token<1, false, 8>("1 wave x12, free, work 8", a, b, n_rec, 11520);
    run_rec_token<1, false, 16>("1 wave x12, free, work 16", a, b, n_rec, 11520);
}

// The beat kernels' memory pattern (512-beat tiles, 16-byte operand loads, one control word and one or two result words per
// beat, results staged through LDS) over separate arrays a / b / ctl / out0 / out1 and over a fused layout in which a tile's
// operands and control words (and its results) are contiguous: would a tiled layout for beats raise their ceiling?
template <bool EUC, bool FUSED>
__global__ __launch_bounds__(256) void k_beatpat(const float *in, float *out, size_t n) {
    constexpr int ROW = EUC ? 16 : 8, LPB = ROW / 4, BPP = 256 / LPB, PASSES = EUC ? 8 : 4, SUB = PASSES * BPP;
    __shared__ float s0[SUB], s1[SUB];
    const int tid = threadIdx.x, part = tid % LPB;
    const size_t tile = blockIdx.x, base = tile * SUB;
    if (base + SUB > n) return;
    const float *a, *b;
    const uint32_t *ctl;
    float *o0, *o1;
    if (FUSED) {
        const float *t = in + tile * (size_t)(2 * SUB * ROW + SUB);
        a = t; b = t + (size_t)SUB * ROW; ctl = reinterpret_cast<const uint32_t *>(t + (size_t)2 * SUB * ROW);
        o0 = out + tile * (size_t)(2 * SUB); o1 = o0 + SUB;
    } else {
        a = in + base * ROW; b = in + n * ROW + base * ROW; ctl = reinterpret_cast<const uint32_t *>(in + 2 * n * ROW) + base;
        o0 = out + base; o1 = out + n + base;
    }
    f4v va[PASSES], vb[PASSES];
    uint32_t vc[PASSES];
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
        const int bl = u * BPP + tid / LPB;
        va[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(a + (size_t)bl * ROW + part * 4));
        vb[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(b + (size_t)bl * ROW + part * 4));
        vc[u] = __builtin_nontemporal_load(ctl + bl);
    }
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
        const int bl = u * BPP + tid / LPB;
        const f4v p = va[u] * vb[u];
        const float r = (p.x + p.y) + (p.z + p.w) + (float)vc[u];
        if (part == 0) { s0[bl] = r; s1[bl] = r * 2.0f; }
    }
    __syncthreads();
    for (int t0 = tid; t0 < SUB; t0 += 256) {
        __builtin_nontemporal_store(s0[t0], o0 + t0);
        if (!EUC) __builtin_nontemporal_store(s1[t0], o1 + t0);
    }
}

template <bool EUC, bool FUSED> static void run_beatpat(const char *name, const void *in, void *out, size_t n, unsigned lds) {
    constexpr int SUB = 512;
    const unsigned blocks = (unsigned)(n / SUB);
    double ms = time_ms([&] { hipLaunchKernelGGL((k_beatpat<EUC, FUSED>), dim3(blocks), dim3(256), lds, 0, (const float *)in, (float *)out, n); });
    const double bytes = (double)n * (EUC ? 136.0 : 76.0);
    printf("%-44s lds %6u  %8.1f GB/s  (%.3f ms)\n", name, lds, bytes / ms / 1e6, ms);
}

static void beat_probes(void *a, void *b) {
    const size_t n = (size_t)1 << 24;
    const unsigned caps[] = {0u, 12800u, 19200u, 26880u, 32000u};  // as many as fit, then 7 / 6 / 5 / 4 workgroups per CU with the 4 KiB of the kernel
    for (unsigned lds : caps) {
        run_beatpat<false, false>("angular pattern, separate arrays", a, b, n, lds);
        run_beatpat<false, true>("angular pattern, fused tiles", a, b, n, lds);
        run_beatpat<true, false>("euclidean pattern, separate arrays", a, b, n, lds);
        run_beatpat<true, true>("euclidean pattern, fused tiles", a, b, n, lds);
    }
}

int main(int argc, char **argv) {
    // second argument: resident 256-thread workgroups per CU (LDS granules of 1280 B, 128 per CU); 0 = whatever fits
    if (argc > 2 && atoi(argv[2]) >= 3) g_lds = (128u / (unsigned)atoi(argv[2])) * 1280u;
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 2048) << 20;  // MiB
    void *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0x11, bytes));
    CK(hipMemset(b, 0x22, bytes));
    CK(hipDeviceSynchronize());
    if (argc > 3 && !strcmp(argv[3], "beat")) {  // the beat kernels' pattern, separate arrays against fused tiles
        beat_probes(a, b);
        CK(hipFree(a));
        CK(hipFree(b));
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "token")) {  // throttled load phases
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_token<12, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_token<12, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_token<12, false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_token<12, true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_token<6, false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_token<6, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_token<6, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_token<6, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_token<6, false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rec_token<6, true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 140000));
        token_probes(a, b, bytes);
        CK(hipFree(a));
        CK(hipFree(b));
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "rec")) {  // only the record-stream pattern
        rec_probes(a, b, bytes);
        CK(hipFree(a));
        CK(hipFree(b));
        return 0;
    }
    run_read<float, false, 1>("read dword 1 stream", a, b, bytes);
    run_read<float, true, 1>("read dword nt 1 stream", a, b, bytes);
    run_read<f2v, false, 1>("read dwordx2 1 stream", a, b, bytes);
    run_read<f2v, true, 1>("read dwordx2 nt 1 stream", a, b, bytes);
    run_read<f4v, false, 1>("read dwordx4 1 stream", a, b, bytes);
    run_read<f4v, true, 1>("read dwordx4 nt 1 stream", a, b, bytes);
    run_read<float, true, 32>("read dword nt 32 streams", a, b, bytes);
    run_read<f2v, true, 32>("read dwordx2 nt 32 streams", a, b, bytes);
    run_read<f4v, true, 32>("read dwordx4 nt 32 streams", a, b, bytes);
    run_read<f4v, false, 32>("read dwordx4 32 streams", a, b, bytes);
    run_copy<float, false>("copy dword", a, b, bytes);
    run_copy<f2v, true>("copy dwordx2 nt", a, b, bytes);
    run_copy<f4v, false>("copy dwordx4", a, b, bytes);
    run_copy<f4v, true>("copy dwordx4 nt", a, b, bytes);
    run_write<float, true, 1>("write dword nt", b, bytes);
    run_write<f2v, true, 1>("write dwordx2 nt", b, bytes);
    run_write<f4v, true, 1>("write dwordx4 nt", b, bytes);
    run_write<f4v, false, 1>("write dwordx4", b, bytes);
    run_write<f2v, true, 5>("write dwordx2 nt 5 streams", b, bytes);
    run_write<f4v, true, 5>("write dwordx4 nt 5 streams", b, bytes);
    run_mix<float, 34, 5, false>("mix 34r+5w dword", a, b, bytes);
    run_mix<f2v, 34, 5, false>("mix 34r+5w dwordx2", a, b, bytes);
    run_mix<f4v, 34, 5, false>("mix 34r+5w dwordx4", a, b, bytes);
    run_mix<f2v, 34, 5, true>("mix 34r x2 + 5w x4 (paired)", a, b, bytes);
    run_mix<f2v, 34, 0, false>("mix 34r+0w dwordx2", a, b, bytes);
    run_mix<f2v, 16, 3, false>("mix 16r+3w dwordx2", a, b, bytes);
    run_mix<f4v, 16, 3, false>("mix 16r+3w dwordx4", a, b, bytes);
    CK(hipFree(a));
    CK(hipFree(b));
    return 0;
}
