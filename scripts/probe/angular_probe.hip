// Throw-away measurement: what the memory pattern of the angular beat alone can reach (MI355X).
// Streams: q[n][8], c[n][8] floats (32 B per beat each), ctl[n] (4 B), dot[n], norm[n] (4 B each) = 76 B per beat.
// No accumulation across beats: every beat writes its own tree sums (results meaningless), so no LDS, no walk.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/probe/angular_probe.hip -o scripts/probe/angular_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

// LPB lanes per beat (2: 16-byte loads, 4: 8-byte loads, 1: two 16-byte loads per operand and lane), U beats per lane group
template <int LPB, int U>
__global__ __launch_bounds__(256) void k_ang(const float *__restrict__ q, const float *__restrict__ c, const uint32_t *__restrict__ ctl,
                                             float *__restrict__ dot, float *__restrict__ norm, uint64_t n) {
    constexpr int BPP = 256 / LPB;                       // beats per pass of the workgroup
    const uint64_t base = (uint64_t)blockIdx.x * (BPP * U);
    const int part = threadIdx.x % LPB, bl = threadIdx.x / LPB;
    float s0[U], s1[U];
    if (LPB == 2) {
        f4v a[U], b[U]; uint32_t m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t beat = base + (uint64_t)u * BPP + bl;
            if (beat < n) {
                a[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(q + beat * 8 + part * 4));
                b[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(c + beat * 8 + part * 4));
                m[u] = __builtin_nontemporal_load(ctl + beat);
            } else { a[u] = (f4v)(0.f); b[u] = (f4v)(0.f); m[u] = 0; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float x = 0.f, y = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const bool on = (m[u] >> (part * 4 + j)) & 1u; const float qq = on ? a[u][j] : 0.f, cc = on ? b[u][j] : 0.f; x += qq * cc; y += cc * cc; }
            x += __shfl_xor(x, 1); y += __shfl_xor(y, 1);
            s0[u] = x; s1[u] = y;
        }
    } else if (LPB == 4) {
        f2v a[U], b[U]; uint32_t m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t beat = base + (uint64_t)u * BPP + bl;
            if (beat < n) {
                a[u] = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(q + beat * 8 + part * 2));
                b[u] = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(c + beat * 8 + part * 2));
                m[u] = __builtin_nontemporal_load(ctl + beat);
            } else { a[u] = (f2v)(0.f); b[u] = (f2v)(0.f); m[u] = 0; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float x = 0.f, y = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) { const bool on = (m[u] >> (part * 2 + j)) & 1u; const float qq = on ? a[u][j] : 0.f, cc = on ? b[u][j] : 0.f; x += qq * cc; y += cc * cc; }
            x += __shfl_xor(x, 1); y += __shfl_xor(y, 1); x += __shfl_xor(x, 2); y += __shfl_xor(y, 2);
            s0[u] = x; s1[u] = y;
        }
    } else {
        f4v a0[U], a1[U], b0[U], b1[U]; uint32_t m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t beat = base + (uint64_t)u * BPP + bl;
            if (beat < n) {
                a0[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(q + beat * 8));
                a1[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(q + beat * 8 + 4));
                b0[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(c + beat * 8));
                b1[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(c + beat * 8 + 4));
                m[u] = __builtin_nontemporal_load(ctl + beat);
            } else { a0[u] = a1[u] = b0[u] = b1[u] = (f4v)(0.f); m[u] = 0; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float x = 0.f, y = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on0 = (m[u] >> j) & 1u, on1 = (m[u] >> (4 + j)) & 1u;
                const float q0 = on0 ? a0[u][j] : 0.f, c0 = on0 ? b0[u][j] : 0.f, q1 = on1 ? a1[u][j] : 0.f, c1 = on1 ? b1[u][j] : 0.f;
                x += q0 * c0; y += c0 * c0; x += q1 * c1; y += c1 * c1;
            }
            s0[u] = x; s1[u] = y;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t beat = base + (uint64_t)u * BPP + bl;
        if (part == 0 && beat < n) {
            __builtin_nontemporal_store(s0[u], dot + beat);
            __builtin_nontemporal_store(s1[u], norm + beat);
        }
    }
}

// packed operands: one array [n][16] = q[0..7] | c[0..7] per beat (64 B contiguous), 4 lanes per beat, 16 B each
template <int U>
__global__ __launch_bounds__(256) void k_ang_packed(const float *__restrict__ qc, const float *__restrict__, const uint32_t *__restrict__ ctl,
                                                    float *__restrict__ dot, float *__restrict__ norm, uint64_t n) {
    constexpr int BPP = 64;
    const uint64_t base = (uint64_t)blockIdx.x * (BPP * U);
    const int part = threadIdx.x & 3, bl = threadIdx.x >> 2;
    f4v a[U]; uint32_t m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t beat = base + (uint64_t)u * BPP + bl;
        if (beat < n) {
            a[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(qc + beat * 16 + part * 4));
            m[u] = __builtin_nontemporal_load(ctl + beat);
        } else { a[u] = (f4v)(0.f); m[u] = 0; }
    }
    float s0[U], s1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // lanes 0, 1 hold q, lanes 2, 3 hold c: fetch the partner across the quad
        f4v o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(a[u][j], 2);
        float x = 0.f, y = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const bool on = (m[u] >> ((part & 1) * 4 + j)) & 1u; const float qq = on ? a[u][j] : 0.f, cc = on ? o[j] : 0.f; x += qq * cc; y += cc * cc; }
        x += __shfl_xor(x, 1); y += __shfl_xor(y, 1);
        s0[u] = x; s1[u] = y;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t beat = base + (uint64_t)u * BPP + bl;
        if (part == 0 && beat < n) { __builtin_nontemporal_store(s0[u], dot + beat); __builtin_nontemporal_store(s1[u], norm + beat); }
    }
}

// the same traffic with coalesced result stores: results go through LDS, then one 4-byte store per lane and output
template <int U>
__global__ __launch_bounds__(256) void k_ang_lds(const float *__restrict__ q, const float *__restrict__ c, const uint32_t *__restrict__ ctl,
                                                 float *__restrict__ dot, float *__restrict__ norm, uint64_t n) {
    constexpr int BPP = 128;
    __shared__ float s_d[BPP * U], s_n[BPP * U];
    const uint64_t base = (uint64_t)blockIdx.x * (BPP * U);
    const int part = threadIdx.x & 1, bl = threadIdx.x >> 1;
    f4v a[U], b[U]; uint32_t m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t beat = base + (uint64_t)u * BPP + bl;
        if (beat < n) {
            a[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(q + beat * 8 + part * 4));
            b[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(c + beat * 8 + part * 4));
            m[u] = __builtin_nontemporal_load(ctl + beat);
        } else { a[u] = (f4v)(0.f); b[u] = (f4v)(0.f); m[u] = 0; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float x = 0.f, y = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const bool on = (m[u] >> (part * 4 + j)) & 1u; const float qq = on ? a[u][j] : 0.f, cc = on ? b[u][j] : 0.f; x += qq * cc; y += cc * cc; }
        x += __shfl_xor(x, 1); y += __shfl_xor(y, 1);
        if (part == 0) { s_d[u * BPP + bl] = x; s_n[u * BPP + bl] = y; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < BPP * U; k += 256) {
        const uint64_t beat = base + k;
        if (beat < n) { __builtin_nontemporal_store(s_d[k], dot + beat); __builtin_nontemporal_store(s_n[k], norm + beat); }
    }
}

__global__ void k_fill(float *p, uint64_t n, uint32_t seed) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h & 0xFFFFFF) / 16777216.0f * 200.0f - 100.0f;
    }
}

static unsigned lds_for(int w, unsigned used) { if (w >= 99) return 0u; const unsigned share = (128u / (unsigned)w) * 1280u; return share > used ? share - used : 0u; }

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
        float *q = nullptr, *c = nullptr, *dot = nullptr, *norm = nullptr; uint32_t *ctl = nullptr;
        CK(hipMalloc(&q, n * 64)); /* q doubles as the packed [n][16] array */ CK(hipMalloc(&c, n * 32)); CK(hipMalloc(&ctl, n * 4)); CK(hipMalloc(&dot, n * 4)); CK(hipMalloc(&norm, n * 4));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, q, n * 16, 1u);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c, n * 8, 2u);
        CK(hipMemset(ctl, 0xFF, n * 4));
        CK(hipDeviceSynchronize());
        const int caps[] = {99, 8, 6, 5, 4, 3};
        for (int round = 0; round < 2; ++round)
        for (int w : caps) {
            double ms;
#define RUN(NAME, K, BEATS_PER_WG, USED) { const unsigned blocks = (unsigned)((n + (BEATS_PER_WG) - 1) / (BEATS_PER_WG)); \
            ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), lds_for(w, USED), 0, q, c, ctl, dot, norm, n); }, 10, 40); \
            printf("n 2^%d cap %2d  %-34s %.4f ms frac %.4f\n", lg, w, NAME, ms, n * 76.0 / ms / 1e6 / 8000.0); }
            RUN("packed [n][16], 2 beats per quad", (k_ang_packed<2>), 128, 0)
            RUN("packed [n][16], 4 beats per quad", (k_ang_packed<4>), 256, 0)
            RUN("packed [n][16], 8 beats per quad", (k_ang_packed<8>), 512, 0)
            RUN("2 lanes/beat, 1 beat per pair", (k_ang<2, 1>), 128, 0)
            RUN("2 lanes/beat, 2 beats per pair", (k_ang<2, 2>), 256, 0)
            RUN("2 lanes/beat, 4 beats per pair", (k_ang<2, 4>), 512, 0)
            RUN("2 lanes/beat, 4 beats, LDS stores", (k_ang_lds<4>), 512, 4096)
            RUN("2 lanes/beat, 2 beats, LDS stores", (k_ang_lds<2>), 256, 2048)
            fflush(stdout);
        }
        CK(hipFree(q)); CK(hipFree(c)); CK(hipFree(ctl)); CK(hipFree(dot)); CK(hipFree(norm));
    }
    return 0;
}
