// micro-benchmark behind the Gaussian sketch's generator budget (round 5): how many VALU instructions of which class fit beside
// a v_mfma_f32_32x32x16_bf16 stream on one SIMD, with one and with two waves per SIMD, and what whole generator candidates cost
// when they are placed (a) as one clump in front of a step's 8 MFMAs (what the product kernel does) or (b) woven between them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scratch/bin/gen_bench scratch/gen_bench.hip && scratch/bin/gen_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// filler classes: N independent instructions on 8 rotating registers
template <int OP> __device__ __forceinline__ void filler(uint32_t (&r)[8], int k, uint32_t b) {
    uint32_t &x = r[k & 7];
    if constexpr (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if constexpr (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if constexpr (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if constexpr (OP == 3) asm volatile("v_log_f32 %0, %0" : "+v"(x));
    else if constexpr (OP == 4) asm volatile("v_sin_f32 %0, %0" : "+v"(x));
    else if constexpr (OP == 5) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x));
    else if constexpr (OP == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if constexpr (OP == 7) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
    else if constexpr (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
}

// ---- A: 8 MFMAs per step, N fillers of class OP behind each MFMA --------------------------------------------------------
template <int OP, int N, int WAVES> __global__ __launch_bounds__(64 * WAVES) void mfma_fill(float *out, int iters, uint32_t seed) {
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    uint32_t r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = seed * (k + 1) + threadIdx.x;
    const u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {seed, seed, seed, seed};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc[t] = mfma(a, b, acc[t]);
#pragma unroll
            for (int k = 0; k < N; ++k) filler<OP>(r, t * N + k, seed);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0];
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) x ^= r[k];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s + static_cast<float>(x & 1);
}

// ---- B: generator candidates ------------------------------------------------------------------------------------------------
struct Key { uint32_t k0, k1; };
template <int ROUNDS> __device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, Key key, uint32_t (&o)[4]) {
    uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0, p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
        const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0, n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
        c1 = static_cast<uint32_t>(p1); c3 = static_cast<uint32_t>(p0); c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
__device__ __forceinline__ uint32_t xoshiro(uint32_t (&s)[4]) {
    const uint32_t result = rotl32(s[0] + s[3], 7) + s[0];
    const uint32_t t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl32(s[3], 11);
    return result;
}
__device__ __forceinline__ uint32_t pack(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t bm16(uint32_t w) {           // Box-Muller pair from 16 + 16 bits
    const float u1 = __builtin_fmaf(static_cast<float>(w & 0xffffu), 1.0f / 65536.0f, 0.5f / 65536.0f);
    const float u2 = static_cast<float>(w >> 16) * (1.0f / 65536.0f);
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    return pack(rad * __builtin_amdgcn_cosf(u2), rad * __builtin_amdgcn_sinf(u2));
}
__device__ __forceinline__ uint32_t bm8(uint32_t h) {            // Box-Muller pair from 16 bits: 10-bit radius, 6-bit angle (+ half a step)
    const float u1 = __builtin_fmaf(static_cast<float>(h & 0x3ffu), 1.0f / 1024.0f, 0.5f / 1024.0f);
    const float u2 = __builtin_fmaf(static_cast<float>((h >> 10) & 0x3fu), 1.0f / 64.0f, 0.5f / 64.0f);
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    return pack(rad * __builtin_amdgcn_cosf(u2), rad * __builtin_amdgcn_sinf(u2));
}
// GEN: 0 constant operand, 1 Philox-10 per fragment + bm16 (round 4), 2 xoshiro128++ x4 + bm16 (round 5), 3 xoshiro x2 + bm8,
//      4 xoshiro x4 only (no Box-Muller), 5 bm16 only (words = counter), 6 Philox-10 per TWO fragments + bm8
//      7 xoshiro x4 + inverse-CDF table in LDS (16384 half-normal bf16 entries, 2 gathers per dword, signs from bits 15 / 31)
struct GenState { uint32_t s0[4], s1[4]; uint32_t ctr; uint32_t w[4]; const uint8_t *table; };
__device__ __forceinline__ uint32_t table_pair(const uint8_t *table, uint32_t w) {
    const uint32_t lo = *reinterpret_cast<const uint16_t *>(table + ((w << 1) & 0x7ffeu));
    const uint32_t hi = *reinterpret_cast<const uint16_t *>(table + ((w >> 15) & 0x7ffeu));
    return (lo | (hi << 16)) ^ (w & 0x80008000u);
}
template <int GEN> __device__ __forceinline__ u32x4 generate(GenState &g, Key key, int step) {
    u32x4 a;
    if constexpr (GEN == 0) { a = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; }
    else if constexpr (GEN == 1) { uint32_t w[4]; philox<10>(threadIdx.x, g.ctr++, 0, 1, key, w); for (int q = 0; q < 4; ++q) a[q] = bm16(w[q]); }
    else if constexpr (GEN == 2) { a[0] = bm16(xoshiro(g.s0)); a[1] = bm16(xoshiro(g.s0)); a[2] = bm16(xoshiro(g.s1)); a[3] = bm16(xoshiro(g.s1)); }
    else if constexpr (GEN == 3) { const uint32_t w0 = xoshiro(g.s0), w1 = xoshiro(g.s1); a[0] = bm8(w0); a[1] = bm8(w0 >> 16); a[2] = bm8(w1); a[3] = bm8(w1 >> 16); }
    else if constexpr (GEN == 4) { a[0] = xoshiro(g.s0); a[1] = xoshiro(g.s0); a[2] = xoshiro(g.s1); a[3] = xoshiro(g.s1); }
    else if constexpr (GEN == 5) { for (int q = 0; q < 4; ++q) a[q] = bm16(g.ctr + q * 0x9E3779B9u); g.ctr += 77; }
    else if constexpr (GEN == 6) {
        if ((step & 1) == 0) philox<10>(threadIdx.x, g.ctr++, 0, 1, key, g.w);
        const uint32_t w0 = g.w[2 * (step & 1)], w1 = g.w[2 * (step & 1) + 1];
        a[0] = bm8(w0); a[1] = bm8(w0 >> 16); a[2] = bm8(w1); a[3] = bm8(w1 >> 16);
    }
    else if constexpr (GEN == 7) {
        a[0] = table_pair(g.table, xoshiro(g.s0)); a[1] = table_pair(g.table, xoshiro(g.s0));
        a[2] = table_pair(g.table, xoshiro(g.s1)); a[3] = table_pair(g.table, xoshiro(g.s1));
    }
    return a;
}
#define GEN_TABLE(G, g, nthreads)                                                                                                \
    __shared__ __attribute__((aligned(16))) uint16_t gtable[(G) == 7 ? 16384 : 8];                                               \
    if constexpr ((G) == 7) { for (int i = threadIdx.x; i < 16384; i += (nthreads)) gtable[i] = static_cast<uint16_t>(0x3c00u + (i >> 4)); } \
    (g).table = reinterpret_cast<const uint8_t *>(gtable);

// the product kernel's shape: per step the fragment, then 8 MFMAs each followed by one conflict-free ds_read_b128 (B operand);
// 8 steps per "stage" and a barrier.  The generator is a clump in front of the step's MFMAs, as hipcc places it in the kernel.
template <int GEN, int WAVES> __global__ __launch_bounds__(64 * WAVES) void gen_clump(float *out, int iters, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 64 * WAVES) reinterpret_cast<uint32_t *>(lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    GenState g;
    GEN_TABLE(GEN, g, 64 * WAVES)
    __syncthreads();
    const Key key{seed, seed ^ 0x5555u};
    philox<10>(threadIdx.x, blockIdx.x, 0, 2, key, g.s0);
    philox<10>(threadIdx.x, blockIdx.x, 1, 2, key, g.s1);
    g.ctr = seed;
    const uint8_t *frag = lds + lane * 16;
    u32x4 bq[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bq[t] = *reinterpret_cast<const u32x4 *>(frag + 1024 * t);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const u32x4 a = generate<GEN>(g, key, ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                acc[t] = mfma(a, bq[t], acc[t]);
                bq[t] = *reinterpret_cast<const u32x4 *>(frag + 1024 * ((t + ks) & 7) + 8192 * (ks & 1));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][7];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

// the same work with the NEXT step's fragment produced between this step's MFMAs (no sched barriers: hipcc interleaves freely)
template <int GEN, int WAVES> __global__ __launch_bounds__(64 * WAVES) void gen_woven(float *out, int iters, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 64 * WAVES) reinterpret_cast<uint32_t *>(lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    GenState g;
    GEN_TABLE(GEN, g, 64 * WAVES)
    __syncthreads();
    const Key key{seed, seed ^ 0x5555u};
    philox<10>(threadIdx.x, blockIdx.x, 0, 2, key, g.s0);
    philox<10>(threadIdx.x, blockIdx.x, 1, 2, key, g.s1);
    g.ctr = seed;
    const uint8_t *frag = lds + lane * 16;
    u32x4 bq[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bq[t] = *reinterpret_cast<const u32x4 *>(frag + 1024 * t);
    u32x4 a = generate<GEN>(g, key, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const u32x4 an = generate<GEN>(g, key, ks + 1);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                acc[t] = mfma(a, bq[t], acc[t]);
                bq[t] = *reinterpret_cast<const u32x4 *>(frag + 1024 * ((t + ks) & 7) + 8192 * (ks & 1));
            }
            a = an;
        }
        __syncthreads();
    }
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][7];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

// wave specialisation: waves 0..3 (one per SIMD) multiply -- 8 MFMAs + 8 ds_read_b128 per step, the A fragment read from LDS --
// while waves 4..7 (their SIMD partners) only generate: the fragments of the NEXT stage (8 steps) for "their" multiply wave,
// written to LDS; one barrier per stage.  If a VALU-only wave and an MFMA-only wave share a SIMD at "both ~ max", the
// generator is hidden behind the matrix pipe.
template <int GEN> __global__ __launch_bounds__(512) void gen_special(float *out, int iters, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[16384 + 2 * 4 * 8 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (16384 + 65536) / 4; i += 512) reinterpret_cast<uint32_t *>(lds)[i] = 0x3c003c00u + (i & 0xff);
    __syncthreads();
    uint8_t *abuf = lds + 16384;
    float s = 0.0f;
    GenState g;
    GEN_TABLE(GEN, g, 512)
    __syncthreads();
    if (wave < 4) {
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
        const uint8_t *frag = lds + lane * 16;
        u32x4 bq[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) bq[t] = *reinterpret_cast<const u32x4 *>(frag + 1024 * t);
        for (int it = 0; it < iters; ++it) {
            const uint8_t *ab = abuf + ((it & 1) * 4 + wave) * 8192 + lane * 16;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const u32x4 a = *reinterpret_cast<const u32x4 *>(ab + 1024 * ks);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[t] = mfma(a, bq[t], acc[t]);
                    bq[t] = *reinterpret_cast<const u32x4 *>(frag + 1024 * ((t + ks) & 7) + 8192 * (ks & 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][7];
    } else {
        const Key key{seed, seed ^ 0x5555u};
        philox<10>(threadIdx.x, blockIdx.x, 0, 2, key, g.s0);
        philox<10>(threadIdx.x, blockIdx.x, 1, 2, key, g.s1);
        g.ctr = seed;
        for (int it = 0; it < iters; ++it) {
            uint8_t *ab = abuf + (((it + 1) & 1) * 4 + (wave - 4)) * 8192 + lane * 16;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) *reinterpret_cast<u32x4 *>(ab + 1024 * ks) = generate<GEN>(g, key, ks);
            __syncthreads();
        }
        s = static_cast<float>(g.s0[0] & 1);
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename K> int timeit(const char *name, K kern, int threads, float *out, double mfmas_per_iter_per_wave) {
    const int blocks = 256, iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 50, 12345u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double waves_per_simd = threads / 64 / 4.0;
    const double ns_per_mfma_per_simd = best * 1e6 / (iters * mfmas_per_iter_per_wave * waves_per_simd);
    const double tflops = 2.0 * 32 * 32 * 16 * iters * mfmas_per_iter_per_wave * (threads / 64) * blocks / (best * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %6.2f ns per MFMA per SIMD  %7.0f TFLOP/s\n", name, best, ns_per_mfma_per_simd, tflops);
    return 0;
}

#define FILL(OP, N, W) timeit("fill op" #OP " n=" #N " waves=" #W, mfma_fill<OP, N, W>, 64 * W, out, 8)
#define FILLN(OP, W) FILL(OP, 0, W); FILL(OP, 2, W); FILL(OP, 4, W); FILL(OP, 6, W); FILL(OP, 8, W); FILL(OP, 12, W); FILL(OP, 16, W)
#define GEN(G, W) timeit("clump gen" #G " waves=" #W, gen_clump<G, W>, 64 * W, out, 64); timeit("woven gen" #G " waves=" #W, gen_woven<G, W>, 64 * W, out, 64)

int main() {
    float *out; CHECK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    printf("# A: v_mfma_f32_32x32x16_bf16 + N fillers behind each (op0 xor, 1 mul_lo, 2 mul_hi, 3 log, 4 sin, 5 alignbit, 6 cvt_pk_bf16, 7 sqrt, 8 fma); 256 workgroups of 4 / 8 waves\n");
    FILLN(0, 8); FILLN(0, 4);
    FILL(1, 4, 8); FILL(1, 8, 8); FILL(2, 4, 8); FILL(2, 8, 8); FILL(3, 4, 8); FILL(3, 8, 8); FILL(4, 4, 8); FILL(4, 8, 8);
    FILL(5, 8, 8); FILL(6, 8, 8); FILL(7, 4, 8); FILL(7, 8, 8); FILL(8, 8, 8); FILL(8, 16, 8);
    printf("# B: generator candidates beside 8 MFMAs + 8 ds_read_b128 per step (gen0 none, 1 Philox-10 + BM16 = round 4, 2 xoshiro x4 + BM16 = round 5, 3 xoshiro x2 + BM8, 4 xoshiro x4 only, 5 BM16 only, 6 Philox-10 per two fragments + BM8, 7 xoshiro x4 + 32 KB inverse-CDF table in LDS)\n");
    GEN(0, 8); GEN(1, 8); GEN(2, 8); GEN(3, 8); GEN(4, 8); GEN(5, 8); GEN(6, 8); GEN(7, 8);
    GEN(0, 4); GEN(1, 4); GEN(2, 4); GEN(3, 4); GEN(7, 4);
    printf("# C: wave specialisation, 4 multiply waves + 4 generator waves per workgroup (ns per MFMA per SIMD counts the 4 multiply waves)\n");
    timeit("special gen0", gen_special<0>, 512, out, 32); timeit("special gen1", gen_special<1>, 512, out, 32); timeit("special gen2", gen_special<2>, 512, out, 32);
    timeit("special gen3", gen_special<3>, 512, out, 32); timeit("special gen5", gen_special<5>, 512, out, 32); timeit("special gen7", gen_special<7>, 512, out, 32);
    return 0;
}
