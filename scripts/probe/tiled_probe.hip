// Throw-away measurement: the Quadbox kernel over a tiled operand layout ([tile][34 fields][T ops] floats in,
// [tile][5][T] words out) against the plain 39-array layout, same arithmetic (box_one), same residency caps.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I raytracer_amd/csrc -I include scripts/probe/tiled_probe.hip -o gpurun_out/tiled_probe
#include "ops.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#pragma clang fp contract(off)
using namespace rtdp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int T, int V>
__global__ __launch_bounds__(256) void k_tiled(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    // T ops per tile; a workgroup covers 256*V consecutive ops = 256*V/T tiles (T <= 256*V) or a part of one
    const uint64_t i = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * V;
    if (i >= n) return;
    const uint64_t tile = i / T, w = i % T;
    const float *p = in + tile * (uint64_t)(kBoxFields * T) + w;
    typedef float FV __attribute__((ext_vector_type(V)));
    typedef uint32_t UV __attribute__((ext_vector_type(V)));
    FV x[kBoxFields];
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f) x[f] = __builtin_nontemporal_load(reinterpret_cast<const FV *>(p + f * T));
    FV to[4]; UV fo;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        float r[kBoxFields], t[4]; uint32_t fl;
#pragma unroll
        for (int f = 0; f < kBoxFields; ++f) r[f] = x[f][v];
        box_one(r, t, fl);
#pragma unroll
        for (int k = 0; k < 4; ++k) to[k][v] = t[k];
        fo[v] = fl;
    }
    uint32_t *q = out + tile * (uint64_t)(5 * T) + w;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(to[k], reinterpret_cast<FV *>(q + k * T));
    __builtin_nontemporal_store(fo, reinterpret_cast<UV *>(q + 4 * T));
}

template <int T>
__global__ __launch_bounds__(256) void k_tiled1(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t tile = i / T, w = i % T;
    const float *p = in + tile * (uint64_t)(kBoxFields * T) + w;
    float x[kBoxFields];
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f) x[f] = __builtin_nontemporal_load(p + f * T);
    float t[4]; uint32_t fl;
    box_one(x, t, fl);
    uint32_t *q = out + tile * (uint64_t)(5 * T) + w;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(__float_as_uint(t[k]), q + k * T);
    __builtin_nontemporal_store(fl, q + 4 * T);
}

template <typename T> __device__ __forceinline__ T ld_sb(const void *uniform_base, uint32_t lane_off) {
    typedef const T __attribute__((address_space(1))) *G;
    uintptr_t b = reinterpret_cast<uintptr_t>(uniform_base);
    asm volatile("" : "+s"(b));
    return __builtin_nontemporal_load((G)b + lane_off);
}
template <typename T> __device__ __forceinline__ void st_sb(void *uniform_base, uint32_t lane_off, T v) {
    typedef T __attribute__((address_space(1))) *G;
    uintptr_t b = reinterpret_cast<uintptr_t>(uniform_base);
    asm volatile("" : "+s"(b));
    __builtin_nontemporal_store(v, (G)b + lane_off);
}
// scalar tile base + one shared lane offset
template <int T>
__global__ __launch_bounds__(256) void k_tiled_s(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t first = (uint64_t)blockIdx.x * 256u;
    if (first + threadIdx.x >= n) return;
    const uint64_t tile = first / T;
    const uint32_t w = (uint32_t)(first % T) + threadIdx.x;
    const float *p = in + tile * (uint64_t)(kBoxFields * T);
    float x[kBoxFields];
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f) x[f] = ld_sb<float>(p + (size_t)f * T, w);
    float t[4]; uint32_t fl;
    box_one(x, t, fl);
    uint32_t *q = out + tile * (uint64_t)(5 * T);
#pragma unroll
    for (int k = 0; k < 4; ++k) st_sb<uint32_t>(q + (size_t)k * T, w, __float_as_uint(t[k]));
    st_sb<uint32_t>(q + (size_t)4 * T, w, fl);
}

// scalar base, with PACE idle cycles (s_sleep units of 64 clocks are too coarse: s_nop) after each load
template <int T, int PACE>
__global__ __launch_bounds__(256) void k_tiled_p(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t first = (uint64_t)blockIdx.x * 256u;
    if (first + threadIdx.x >= n) return;
    const uint64_t tile = first / T;
    const uint32_t w = (uint32_t)(first % T) + threadIdx.x;
    const float *p = in + tile * (uint64_t)(kBoxFields * T);
    float x[kBoxFields];
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f) {
        x[f] = ld_sb<float>(p + (size_t)f * T, w);
        if (PACE == 1) asm volatile("s_nop 3");
        if (PACE == 2) asm volatile("s_nop 7");
        if (PACE == 3) asm volatile("s_nop 15");
        if (PACE == 4) asm volatile("s_nop 15\n s_nop 15");
    }
    float t[4]; uint32_t fl;
    box_one(x, t, fl);
    uint32_t *q = out + tile * (uint64_t)(5 * T);
#pragma unroll
    for (int k = 0; k < 4; ++k) st_sb<uint32_t>(q + (size_t)k * T, w, __float_as_uint(t[k]));
    st_sb<uint32_t>(q + (size_t)4 * T, w, fl);
}
template <int T>
__global__ __launch_bounds__(256) void k_tri_tiled_s(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t first = (uint64_t)blockIdx.x * 256u;
    if (first + threadIdx.x >= n) return;
    const uint64_t tile = first / T;
    const uint32_t w = (uint32_t)(first % T) + threadIdx.x;
    const float *p = in + tile * (uint64_t)(16 * T);
    float x[kTriFields];
#pragma unroll
    for (int f = 0; f < kTriFields; ++f) x[f] = ld_sb<float>(p + (size_t)f * T, w);
    const uint32_t kk = ld_sb<uint32_t>(p + (size_t)15 * T, w) & 0x3Fu;
    float tn, td; uint32_t h;
    tri_one(x, kk, tn, td, h);
    uint32_t *q = out + tile * (uint64_t)(3 * T);
    st_sb<uint32_t>(q, w, __float_as_uint(tn));
    st_sb<uint32_t>(q + (size_t)T, w, __float_as_uint(td));
    st_sb<uint32_t>(q + (size_t)2 * T, w, h);
}

// vector addressing with PACE idle cycles after each load
template <int T, int PACE>
__global__ __launch_bounds__(256) void k_tiled_vp(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t tile = i / T, w = i % T;
    const float *p = in + tile * (uint64_t)(kBoxFields * T) + w;
    float x[kBoxFields];
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f) {
        x[f] = __builtin_nontemporal_load(p + f * T);
        if (PACE > 0) asm volatile("s_nop %0" :: "n"(PACE - 1));
    }
    float t[4]; uint32_t fl;
    box_one(x, t, fl);
    uint32_t *q = out + tile * (uint64_t)(5 * T) + w;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(__float_as_uint(t[k]), q + k * T);
    __builtin_nontemporal_store(fl, q + 4 * T);
}

template <int T, int PACE>
__global__ __launch_bounds__(256) void k_tri_tiled_vp(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t tile = i / T, w = i % T;
    const float *p = in + tile * (uint64_t)(16 * T) + w;
    float x[kTriFields];
#pragma unroll
    for (int f = 0; f < kTriFields; ++f) {
        x[f] = __builtin_nontemporal_load(p + f * T);
        if (PACE > 0) asm volatile("s_nop %0" :: "n"(PACE - 1));
    }
    const uint32_t kk = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p + 15 * T)) & 0x3Fu;
    float tn, td; uint32_t h;
    tri_one(x, kk, tn, td, h);
    uint32_t *q = out + tile * (uint64_t)(3 * T) + w;
    __builtin_nontemporal_store(__float_as_uint(tn), q);
    __builtin_nontemporal_store(__float_as_uint(td), q + T);
    __builtin_nontemporal_store(h, q + 2 * T);
}

// vector addressing; SPLIT > 0: wait for the first SPLIT loads before issuing the rest; NOP > 0: s_nop builtin after each load
template <int T, int SPLIT, int NOP>
__global__ __launch_bounds__(256) void k_tiled_x(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t tile = i / T, w = i % T;
    const float *p = in + tile * (uint64_t)(kBoxFields * T) + w;
    float x[kBoxFields];
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f) {
        x[f] = __builtin_nontemporal_load(p + f * T);
        if (NOP > 0) asm volatile("s_nop %0" :: "n"(NOP - 1));
        if (SPLIT > 0 && f == SPLIT - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float t[4]; uint32_t fl;
    box_one(x, t, fl);
    uint32_t *q = out + tile * (uint64_t)(5 * T) + w;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(__float_as_uint(t[k]), q + k * T);
    __builtin_nontemporal_store(fl, q + 4 * T);
}

// tile stride (words per op slot in the buffer) as a parameter: WIN words per op are read, SIN words are reserved per tile
template <int T, int SIN, int SOUT>
__global__ __launch_bounds__(256) void k_tri_stride(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t tile = i / T, w = i % T;
    const float *p = in + tile * (uint64_t)(SIN * T) + w;
    float x[kTriFields];
#pragma unroll
    for (int f = 0; f < kTriFields; ++f) x[f] = __builtin_nontemporal_load(p + f * T);
    const uint32_t kk = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p + 15 * T)) & 0x3Fu;
    float tn, td; uint32_t h;
    tri_one(x, kk, tn, td, h);
    uint32_t *q = out + tile * (uint64_t)(SOUT * T) + w;
    __builtin_nontemporal_store(__float_as_uint(tn), q);
    __builtin_nontemporal_store(__float_as_uint(td), q + T);
    __builtin_nontemporal_store(h, q + 2 * T);
}
template <int T, int SIN, int SOUT>
__global__ __launch_bounds__(256) void k_box_stride(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t tile = i / T, w = i % T;
    const float *p = in + tile * (uint64_t)(SIN * T) + w;
    float x[kBoxFields];
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f) {
        x[f] = __builtin_nontemporal_load(p + f * T);
        asm volatile("s_nop 5");
    }
    float t[4]; uint32_t fl;
    box_one(x, t, fl);
    uint32_t *q = out + tile * (uint64_t)(SOUT * T) + w;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(__float_as_uint(t[k]), q + k * T);
    __builtin_nontemporal_store(fl, q + 4 * T);
}

// vector addressing, paced loads; the results of the workgroup's 256 ops go through LDS and out as 16-byte stores (one wave-instruction
// writes 1 KiB of one result word's run instead of 256 B)
template <int T>
__global__ __launch_bounds__(256) void k_tiled_wide_st(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    __shared__ __align__(16) uint32_t s_res[5 * 256];
    const uint64_t first = (uint64_t)blockIdx.x * 256u;
    const uint64_t i = first + threadIdx.x;
    const uint64_t tile = first / T;
    const uint32_t w0 = (uint32_t)(first % T);
    if (i < n) {
        const float *p = in + tile * (uint64_t)(kBoxFields * T) + w0 + threadIdx.x;
        float x[kBoxFields];
#pragma unroll
        for (int f = 0; f < kBoxFields; ++f) {
            x[f] = __builtin_nontemporal_load(p + f * T);
            asm volatile("s_nop 5");
        }
        float t[4]; uint32_t fl;
        box_one(x, t, fl);
#pragma unroll
        for (int k = 0; k < 4; ++k) s_res[k * 256 + threadIdx.x] = __float_as_uint(t[k]);
        s_res[4 * 256 + threadIdx.x] = fl;
    }
    __syncthreads();
    if (first + 256 <= n) {  // full workgroup: 320 chunks of 16 bytes
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        uint32_t *q = out + tile * (uint64_t)(5 * T) + w0;
        for (int c = threadIdx.x; c < 320; c += 256) {
            const int k = c / 64, j = c % 64;   // result word, chunk inside the 1 KiB run
            __builtin_nontemporal_store(reinterpret_cast<const u4 *>(s_res + k * 256)[j], reinterpret_cast<u4 *>(q + (size_t)k * T) + j);
        }
    } else if (i < n) {
        uint32_t *q = out + tile * (uint64_t)(5 * T) + w0 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 5; ++k) q[(size_t)k * T] = s_res[k * 256 + threadIdx.x];
    }
}

struct Ptrs { const float *in[kBoxFields]; uint32_t *out[5]; };
__global__ __launch_bounds__(256) void k_plain(Ptrs a, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float x[kBoxFields];
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f) x[f] = __builtin_nontemporal_load(a.in[f] + i);
    float t[4]; uint32_t fl;
    box_one(x, t, fl);
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(__float_as_uint(t[k]), a.out[k] + i);
    __builtin_nontemporal_store(fl, a.out[4] + i);
}

template <int T>
__global__ __launch_bounds__(256) void k_tri_tiled(const float *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t tile = i / T, w = i % T;
    const float *p = in + tile * (uint64_t)(16 * T) + w;
    float x[kTriFields];
#pragma unroll
    for (int f = 0; f < kTriFields; ++f) x[f] = __builtin_nontemporal_load(p + f * T);
    const uint32_t kk = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p + 15 * T)) & 0x3Fu;
    float tn, td; uint32_t h;
    tri_one(x, kk, tn, td, h);
    uint32_t *q = out + tile * (uint64_t)(3 * T) + w;
    __builtin_nontemporal_store(__float_as_uint(tn), q);
    __builtin_nontemporal_store(__float_as_uint(td), q + T);
    __builtin_nontemporal_store(h, q + 2 * T);
}
struct TPtrs { const float *in[16]; uint32_t *out[3]; };
__global__ __launch_bounds__(256) void k_tri_plain(TPtrs a, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float x[kTriFields];
#pragma unroll
    for (int f = 0; f < kTriFields; ++f) x[f] = __builtin_nontemporal_load(a.in[f] + i);
    const uint32_t kk = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(a.in[15] + i)) & 0x3Fu;
    float tn, td; uint32_t h;
    tri_one(x, kk, tn, td, h);
    __builtin_nontemporal_store(__float_as_uint(tn), a.out[0] + i);
    __builtin_nontemporal_store(__float_as_uint(td), a.out[1] + i);
    __builtin_nontemporal_store(h, a.out[2] + i);
}

template <int PACE>
__global__ __launch_bounds__(256) void k_plain_p(Ptrs a, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float x[kBoxFields];
#pragma unroll
    for (int f = 0; f < kBoxFields; ++f) {
        x[f] = __builtin_nontemporal_load(a.in[f] + i);
        if (PACE > 0) asm volatile("s_nop %0" :: "n"(PACE - 1));
    }
    float t[4]; uint32_t fl;
    box_one(x, t, fl);
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(__float_as_uint(t[k]), a.out[k] + i);
    __builtin_nontemporal_store(fl, a.out[4] + i);
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
    if (sizes.empty()) sizes = {22, 24, 26};
    for (int lg : sizes) {
        const uint64_t n = 1ull << lg;
        float *in; uint32_t *out;
        const uint64_t padded = n + 32768;  // whole tiles for every tile size up to 32768 ops
        CK(hipMalloc(&in, padded * kBoxFields * 4)); CK(hipMalloc(&out, padded * 5 * 4));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, padded * kBoxFields, 12345u);
        CK(hipDeviceSynchronize());
        const unsigned blocks = (unsigned)((n + 255) / 256);
        const int caps[] = {3};
        for (int round = 0; round < 2; ++round)
        for (int w : caps) {
            const unsigned lds = lds_for(w);
            double ms;
#define BOXRUN(NAME, K) ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), lds, 0, in, out, n); }, 10, 40); \
            printf("box n 2^%d cap %2d  %-20s %.4f ms frac %.4f\n", lg, w, NAME, ms, n * 156.0 / ms / 1e6 / 8000.0);
            {
                double ms2;
#define BRUN(NAME, K) ms2 = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), lds_for(w) > 5120 ? lds_for(w) - 5120 : 0, 0, in, out, n); }, 10, 40); \
                printf("box n 2^%d cap %2d  %-28s %.4f ms frac %.4f\n", lg, w, NAME, ms2, n * 156.0 / ms2 / 1e6 / 8000.0);
                ms2 = time_ms([&] { hipLaunchKernelGGL((k_tiled_vp<8192, 6>), dim3(blocks), dim3(256), lds_for(w), 0, in, out, n); }, 10, 40);
                printf("box n 2^%d cap %2d  %-28s %.4f ms frac %.4f\n", lg, w, "4-byte stores", ms2, n * 156.0 / ms2 / 1e6 / 8000.0);
                BRUN("16-byte stores through LDS", (k_tiled_wide_st<8192>))
            }
            fflush(stdout);
        }
        CK(hipFree(in)); CK(hipFree(out));
    }
    return 0;
}
