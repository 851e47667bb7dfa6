// VALU-only cost of the forward's per-group work (bucket+pack, gelu), no memory traffic
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../fewbit_amd/csrc/fewbit_codepack.h"
#include "../fewbit_amd/csrc/fewbit_device.h"
using namespace fewbit_hip;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// plain C++ bucket: binary search, bits materialised with selects of inline constants, no v_addc
__device__ __forceinline__ uint32_t code_k3(const float (&b)[7], float x) {
    const bool A = !(b[3] >= x);
    const float t = A ? b[5] : b[1];
    const bool B = !(t >= x);
    const float p = A ? b[4] : b[0];
    const float q = A ? b[6] : b[2];
    const float t0 = B ? q : p;
    const bool C = !(t0 >= x);
    return (A ? 4u : 0u) | (B ? 2u : 0u) | (C ? 1u : 0u);
}

template <int MODE> __global__ __launch_bounds__(256, 8) void k(float* out, const float* bp, int iters) {
    float b[7];
    for (int j = 0; j < 7; ++j) b[j] = bp[j];
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (threadIdx.x * 8 + i) * 0.001f - 1.0f;
    uint32_t acc = 0; float facc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = v[i] + facc * 1e-9f + it * 1e-6f;
        if (MODE == 32 || MODE == 33 || MODE == 34 || MODE == 35) {
            for (int i = 0; i < 8; ++i) {
                const float a = __builtin_fabsf(x[i]);
                float r = -2.834918860e-06f;
                r = __builtin_fmaf(r, a, 3.937771180e-05f); r = __builtin_fmaf(r, a, -1.861801138e-04f); r = __builtin_fmaf(r, a, -1.369371021e-04f);
                r = __builtin_fmaf(r, a, 7.063421421e-03f); r = __builtin_fmaf(r, a, -5.249617994e-02f); r = __builtin_fmaf(r, a, -4.592081904e-01f);
                r = __builtin_fmaf(r, a, -1.151105165e+00f);
                const float q = __builtin_fmaf(r, a, -1.0f);
                float h = q;
                if (MODE == 33 || MODE == 35) h = __builtin_amdgcn_exp2f(q);
                float rl = x[i];
                if (MODE == 34 || MODE == 35) rl = relu_raw(x[i]);
                x[i] = __builtin_fmaf(-a, h, rl);
            }
        } else if (MODE == 36) {   // gelu with the 8 exps issued back to back
            float q[8], a[8];
            for (int i = 0; i < 8; ++i) {
                a[i] = __builtin_fabsf(x[i]);
                float r = -2.834918860e-06f;
                r = __builtin_fmaf(r, a[i], 3.937771180e-05f); r = __builtin_fmaf(r, a[i], -1.861801138e-04f); r = __builtin_fmaf(r, a[i], -1.369371021e-04f);
                r = __builtin_fmaf(r, a[i], 7.063421421e-03f); r = __builtin_fmaf(r, a[i], -5.249617994e-02f); r = __builtin_fmaf(r, a[i], -4.592081904e-01f);
                r = __builtin_fmaf(r, a[i], -1.151105165e+00f);
                q[i] = __builtin_fmaf(r, a[i], -1.0f);
            }
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n s_nop 0"
                         : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]));
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(-a[i], q[i], relu_raw(x[i]));
        } else if (MODE == 40 || MODE == 41) {
            uint32_t w = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                w |= code_k3(b, x[i]) << (3 * i);
                if (MODE == 41) x[i] = gelu_fast(x[i]);
            }
            acc += w;
        } else if (MODE == 16) {  // interleaved per pair: bucket(pair) then gelu(pair)
            uint32_t w = 0;
#pragma unroll
            for (int i = 7; i > 0; i -= 2) {
                w = push2_k3(w, x[i], x[i - 1], b);
                x[i] = gelu_fast(x[i]);
                x[i - 1] = gelu_fast(x[i - 1]);
            }
            acc += w;
        } else if (MODE == 17) {  // software-skewed: bucket(pair p) with gelu(pair p+1)
            uint32_t w = 0;
            float g7 = gelu_fast(x[7]), g6 = gelu_fast(x[6]);
            w = push2_k3(w, x[7], x[6], b); float g5 = gelu_fast(x[5]), g4 = gelu_fast(x[4]);
            w = push2_k3(w, x[5], x[4], b); float g3 = gelu_fast(x[3]), g2 = gelu_fast(x[2]);
            w = push2_k3(w, x[3], x[2], b); float g1 = gelu_fast(x[1]), g0 = gelu_fast(x[0]);
            w = push2_k3(w, x[1], x[0], b);
            x[7]=g7;x[6]=g6;x[5]=g5;x[4]=g4;x[3]=g3;x[2]=g2;x[1]=g1;x[0]=g0;
            acc += w;
        } else {
        if (MODE & 1) acc += pack_group<3>(x, b);
        if (MODE & 2) { for (int i = 0; i < 8; ++i) x[i] = gelu_fast(x[i]); }
        }
        if (MODE & 4) { uint32_t w = 0; for (int i = 0; i < 4; ++i) { f32x2 f = {x[2*i], x[2*i+1]}; w ^= __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2)); } acc ^= w; }
        for (int i = 0; i < 8; ++i) facc += x[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = facc + acc;
}
template <int MODE> int run(const char* name, float* out, float* bp) {
    const int blocks = 256 * 8, iters = 512;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, bp, 4); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, bp, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double elems = (double)blocks * 256 * iters * 8;
    printf("%-28s %7.3f ms  %.3f ns per wave-element per CU  => 4096^2 elements would take %.2f us\n", name, ms, ms * 1e6 / (elems / 64 / 256), ms * 1e3 * (16777216.0 / elems));
    return 0;
}
int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    float hb[7] = {-2.4f, -0.71f, -0.326f, 1e-4f, 0.326f, 0.71f, 2.41f}; float* bp; CHECK(hipMalloc(&bp, 28)); CHECK(hipMemcpy(bp, hb, 28, hipMemcpyHostToDevice));
    run<0>("overhead only (8 adds)", out, bp); run<1>("bucket+pack k3", out, bp); run<2>("gelu_fast", out, bp); run<3>("bucket+pack + gelu", out, bp); run<7>("bucket+pack + gelu + cvt", out, bp); run<32>("gelu: poly only (9 fma)", out, bp); run<33>("gelu: poly + exp", out, bp); run<34>("gelu: poly + max", out, bp); run<35>("gelu: poly+exp+max", out, bp); run<36>("gelu: exps batched", out, bp); run<40>("c++ bucket (no addc)", out, bp); run<41>("c++ bucket + gelu", out, bp); run<16>("interleaved per pair", out, bp); run<17>("skewed per pair", out, bp);
    return 0;
}
