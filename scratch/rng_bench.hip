// micro-benchmark for the sketch kernel's random-operand generation on gfx950: integer multiplies, transcendentals,
// whole Philox4x32-R calls and Box-Muller pairs, as wave-instructions (or calls) per ns per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int R> __device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

template <int OP> __global__ __launch_bounds__(256) void bench(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f + 3.0f;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == 0) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 2) {
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (OP == 4) {
#define X(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (OP == 5) {
#define X(i) asm volatile("v_cos_f32 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (OP == 6) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 7) {   // v_mad_u64_u32: 64-bit product in one instruction
                uint64_t p0 = __builtin_bit_cast(uint32_t, a0), p1 = __builtin_bit_cast(uint32_t, a1);
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(a0), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(a1), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(a2), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(a3), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(a4), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(a5), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(a6), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(a7), "v"(b) : "vcc");
                acc ^= (uint32_t)p0 ^ (uint32_t)(p1 >> 32);
            }
        }
        if (OP == 10 || OP == 11) {        // 8 Philox calls per iteration (counted as 64 "instructions" below -> divide by 8)
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (OP == 10) philox4x32<10>(it, threadIdx.x, blockIdx.x, j, 1234u + acc, 5678u, o);
                else philox4x32<7>(it, threadIdx.x, blockIdx.x, j, 1234u + acc, 5678u, o);
                acc ^= o[0] ^ o[1] ^ o[2] ^ o[3];
            }
        }
        if (OP == 12) {                    // 8 x 4 Box-Muller pairs from 8 x 128 given bits (no Philox): 64 normals
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t w[4] = {acc + j, acc * 3u + it, acc ^ 0x9E3779B9u, acc + threadIdx.x};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float u1 = (static_cast<float>(w[q] & 0xffffu) + 0.5f) * (1.0f / 65536.0f);
                    const float u2 = static_cast<float>(w[q] >> 16) * (1.0f / 65536.0f);
                    const float rad = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // -2 ln u = -2 ln2 log2 u
                    const float s = rad * __builtin_amdgcn_sinf(u2), c = rad * __builtin_amdgcn_cosf(u2);
                    acc += __builtin_bit_cast(uint32_t, s) ^ __builtin_bit_cast(uint32_t, c);
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(acc & 1);
}

template <int OP> int run(const char* name, float* out, double per_iter) {
    const int blocks = 256 * 8, iters = 1000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double units = (double)blocks * 4 * iters * per_iter;            // per wave
    const double per_cu_per_ns = units / 256 / (ms * 1e6);
    // one CU = 4 SIMDs; cycles (at 2.4 GHz) per unit per SIMD
    printf("%-34s %8.3f ms  units/ns/CU = %.4f  => %.1f cycles per unit per SIMD (2.4 GHz)\n", name, ms, per_cu_per_ns, 4.0 * 2.4 / per_cu_per_ns);
    return 0;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    run<6>("v_fma_f32", out, 64); run<0>("v_mul_hi_u32", out, 64); run<1>("v_mul_lo_u32", out, 64); run<7>("v_mad_u64_u32", out, 64);
    run<2>("v_log_f32", out, 64); run<3>("v_sqrt_f32", out, 64); run<4>("v_sin_f32", out, 64); run<5>("v_cos_f32", out, 64);
    run<10>("philox4x32-10 call (128 bits)", out, 8); run<11>("philox4x32-7 call (128 bits)", out, 8);
    run<12>("box-muller pair (2 normals)", out, 32);
    return 0;
}
