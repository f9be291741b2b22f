// Throw-away measurement: what the memory pattern of the Euclidean beat alone can reach (MI355X).
// Streams: a[n][16], b[n][16] floats (64 B per beat each), ctl[n] (4 B), acc[n] (4 B) = 136 B per beat.
// No accumulation across beats: every beat writes its own tree sum (results meaningless), so no LDS, no walk.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/probe/euclid_probe.hip -o scripts/probe/euclid_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

// 4 lanes per beat, 16-byte loads, U beats per quad; PACE idle cycles behind each load pair
template <int U, int PACE>
__global__ __launch_bounds__(256) void k_euc(const float *__restrict__ a, const float *__restrict__ b, const uint32_t *__restrict__ ctl,
                                             float *__restrict__ acc, uint64_t n) {
    constexpr int BPP = 64;
    const uint64_t base = (uint64_t)blockIdx.x * (BPP * U);
    const int part = threadIdx.x & 3, bl = threadIdx.x >> 2;
    f4v va[U], vb[U]; uint32_t m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t beat = base + (uint64_t)u * BPP + bl;
        if (beat < n) {
            va[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(a + beat * 16 + part * 4));
            vb[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(b + beat * 16 + part * 4));
            m[u] = __builtin_nontemporal_load(ctl + beat);
        } else { va[u] = (f4v)(0.f); vb[u] = (f4v)(0.f); m[u] = 0; }
        if (PACE > 0) asm volatile("s_nop %0" ::"n"(PACE > 0 ? PACE - 1 : 0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float x = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const bool on = (m[u] >> (part * 4 + j)) & 1u; const float d = on ? (va[u][j] - vb[u][j]) : 0.f; x += d * d; }
        x += __shfl_xor(x, 1); x += __shfl_xor(x, 2);
        const uint64_t beat = base + (uint64_t)u * BPP + bl;
        if (part == 0 && beat < n) __builtin_nontemporal_store(x, acc + beat);
    }
}

// LPB lanes per beat with 16 / LPB elements each: LPB = 16 -> 4-byte loads (a wave-instruction covers 256 contiguous bytes, as in
// the box kernel), LPB = 8 -> 8-byte loads; U beats per lane group
template <int LPB, int U>
__global__ __launch_bounds__(256) void k_euc_n(const float *__restrict__ a, const float *__restrict__ b, const uint32_t *__restrict__ ctl,
                                               float *__restrict__ acc, uint64_t n) {
    constexpr int EPL = 16 / LPB;
    constexpr int BPP = 256 / LPB;
    typedef float VT __attribute__((ext_vector_type(EPL)));
    const uint64_t base = (uint64_t)blockIdx.x * (BPP * U);
    const int part = threadIdx.x % LPB, bl = threadIdx.x / LPB;
    float va[U][EPL], vb[U][EPL]; uint32_t m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t beat = base + (uint64_t)u * BPP + bl;
        if (beat < n) {
            if (EPL == 1) {
                va[u][0] = __builtin_nontemporal_load(a + beat * 16 + part);
                vb[u][0] = __builtin_nontemporal_load(b + beat * 16 + part);
            } else {
                const VT x = __builtin_nontemporal_load(reinterpret_cast<const VT *>(a + beat * 16 + part * EPL));
                const VT y = __builtin_nontemporal_load(reinterpret_cast<const VT *>(b + beat * 16 + part * EPL));
#pragma unroll
                for (int j = 0; j < EPL; ++j) { va[u][j] = x[j]; vb[u][j] = y[j]; }
            }
            m[u] = __builtin_nontemporal_load(ctl + beat);
        } else {
#pragma unroll
            for (int j = 0; j < EPL; ++j) { va[u][j] = 0.f; vb[u][j] = 0.f; }
            m[u] = 0;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float x = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) { const bool on = (m[u] >> (part * EPL + j)) & 1u; const float d = on ? (va[u][j] - vb[u][j]) : 0.f; x += d * d; }
#pragma unroll
        for (int o = 1; o < LPB; o <<= 1) x += __shfl_xor(x, o);
        const uint64_t beat = base + (uint64_t)u * BPP + bl;
        if (part == 0 && beat < n) __builtin_nontemporal_store(x, acc + beat);
    }
}

__global__ void k_fill(float *p, uint64_t n, uint32_t seed) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h & 0xFFFFFF) / 16777216.0f * 200.0f - 100.0f;
    }
}

static unsigned lds_for(int w) { return w >= 99 ? 0u : (128u / (unsigned)w) * 1280u; }

template <typename F> static double time_ms(F launch, int warm, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < warm; ++i) launch();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char **argv) {
    std::vector<int> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {24};
    for (int lg : sizes) {
        const uint64_t n = 1ull << lg;
        float *a = nullptr, *b = nullptr, *acc = nullptr; uint32_t *ctl = nullptr;
        CK(hipMalloc(&a, n * 64));
        CK(hipMalloc(&b, n * 64));
        CK(hipMalloc(&ctl, n * 4));
        CK(hipMalloc(&acc, n * 4));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, n * 16, 1u);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, b, n * 16, 2u);
        CK(hipMemset(ctl, 0xFF, n * 4));
        CK(hipDeviceSynchronize());
        const int caps[] = {99, 5, 4, 3};
        for (int round = 0; round < 2; ++round)
        for (int w : caps) {
            double ms;
#define RUN(NAME, K, BEATS_PER_WG) { const unsigned blocks = (unsigned)((n + (BEATS_PER_WG) - 1) / (BEATS_PER_WG)); \
            ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), lds_for(w), 0, a, b, ctl, acc, n); }, 10, 40); \
            printf("n 2^%d cap %2d  %-34s %.4f ms frac %.4f\n", lg, w, NAME, ms, n * 136.0 / ms / 1e6 / 8000.0); }
            RUN("1 beat per quad", (k_euc<1, 0>), 64)
            RUN("2 beats per quad", (k_euc<2, 0>), 128)
            RUN("4 beats per quad", (k_euc<4, 0>), 256)
            RUN("16 lanes/beat (4 B loads), 4 beats", (k_euc_n<16, 4>), 64)
            RUN("16 lanes/beat (4 B loads), 8 beats", (k_euc_n<16, 8>), 128)
            RUN("16 lanes/beat (4 B loads), 16 beats", (k_euc_n<16, 16>), 256)
            RUN("16 lanes/beat (4 B loads), 32 beats", (k_euc_n<16, 32>), 512)
            RUN("8 lanes/beat (8 B loads), 4 beats", (k_euc_n<8, 4>), 128)
            RUN("8 lanes/beat (8 B loads), 8 beats", (k_euc_n<8, 8>), 256)
            RUN("8 lanes/beat (8 B loads), 16 beats", (k_euc_n<8, 16>), 512)
            fflush(stdout);
        }
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(ctl)); CK(hipFree(acc));
    }
    return 0;
}
