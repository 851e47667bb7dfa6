// micro-benchmark: per-instruction VALU throughput on gfx950 (which ops are "1 slot", which cost more)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP> __global__ __launch_bounds__(256) void bench(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, seed * 0.125f)));
    uint64_t m = (uint64_t)iters * 3, m2 = (uint64_t)iters * 5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == 0) {  // v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 1) {  // v_pk_fma_f32 on register pairs
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a0) : "v"(*(double*)&a2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a4) : "v"(*(double*)&a6));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a0) : "v"(*(double*)&a2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a4) : "v"(*(double*)&a6));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a0) : "v"(*(double*)&a2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a4) : "v"(*(double*)&a6));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a0) : "v"(*(double*)&a2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a4) : "v"(*(double*)&a6));
            } else if (OP == 2) {  // v_exp_f32
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (OP == 3) {  // v_cmp_nge_f32 -> sgpr
#define X(i) asm volatile("v_cmp_nge_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a##i), "v"(b));
                REP8(X)
#undef X
            } else if (OP == 4) {  // v_cndmask with sgpr mask
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "s"(m));
                REP8(X)
#undef X
            } else if (OP == 5) {  // v_addc
#define X(i) asm volatile("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(a##i) : "s"(m) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 6) {  // v_cvt_pk_bf16_f32
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 7) {  // v_rcp_f32
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (OP == 8) {  // v_lshl_or_b32
#define X(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 9) {  // v_max_f32
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 10) {  // v_bfe_u32
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 3" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (OP == 11) {  // SALU s_or_b64 x8 (scalar throughput)
#define X(i) asm volatile("s_or_b64 %0, %0, %1" : "+s"(m) : "s"(m2) : "scc");
                REP8(X)
#undef X
            } else if (OP == 12) {  // mixed: 1 VALU fma + 1 SALU
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n s_or_b64 %1, %1, %4" : "+v"(a##i), "+s"(m) : "v"(b), "v"(c), "s"(m2) : "scc");
                REP8(X)
#undef X
            } else if (OP == 13) {  // v_mul_f32
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 14) {  // v_add_f32
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 15) {  // v_pk_add_u16
#define X(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 16) {  // v_lshlrev_b32
#define X(i) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (OP == 17) {  // v_cvt_f32_f16
#define X(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (OP == 18) {  // v_and_b32
#define X(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 19) {  // v_fma with abs/neg modifiers (VOP3)
#define X(i) asm volatile("v_fma_f32 %0, -|%0|, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 20) {  // v_fmac_f32 (VOP2)
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 21) {  // v_perm_b32
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 22) {  // v_mov_b32 dpp quad_perm
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (OP == 23) {  // v_cmp -> vcc (VOPC e32)
#define X(i) asm volatile("v_cmp_nge_f32 vcc, %0, %1" : : "v"(a##i), "v"(b) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 30) {  // mix: fma + cndmask alternating (different registers)
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_cndmask_b32_e64 %1, %1, %2, %4" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c), "s"(m));
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_cndmask_b32_e64 %1, %1, %2, %4" : "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "s"(m));
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_cndmask_b32_e64 %1, %1, %2, %4" : "+v"(a4), "+v"(a5) : "v"(b), "v"(c), "s"(m));
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_cndmask_b32_e64 %1, %1, %2, %4" : "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(m));
            } else if (OP == 31) {  // mix: exp + fma
                asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
                asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3" : "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
                asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3" : "+v"(a4), "+v"(a5) : "v"(b), "v"(c));
                asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3" : "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (OP == 32) {  // mix: exp + cndmask
                asm volatile("v_exp_f32 %0, %0\n v_cndmask_b32_e64 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b), "s"(m));
                asm volatile("v_exp_f32 %0, %0\n v_cndmask_b32_e64 %1, %1, %2, %3" : "+v"(a2), "+v"(a3) : "v"(b), "s"(m));
                asm volatile("v_exp_f32 %0, %0\n v_cndmask_b32_e64 %1, %1, %2, %3" : "+v"(a4), "+v"(a5) : "v"(b), "s"(m));
                asm volatile("v_exp_f32 %0, %0\n v_cndmask_b32_e64 %1, %1, %2, %3" : "+v"(a6), "+v"(a7) : "v"(b), "s"(m));
            } else if (OP == 33) {  // exp + 3 fma
                asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
                asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define X(i) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a##i) : "v"(b));
#undef X
            } else if (OP == 34) {
#define X(i) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 35) {
#define X(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 36) {
#define X(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 37) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 38) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 39) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1 clamp" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 40) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 41) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a##i) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 42) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 43) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (OP == 44) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 45) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 46) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(a##i) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 47) {
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 50) {  // clustered: 16 cndmask then 16 fma
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "s"(m));
                REP8(X) REP8(X)
#undef X
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
                REP8(X) REP8(X)
#undef X
            } else if (OP == 51) {  // alternating: 16 x (cndmask, fma)
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %3\n v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c), "s"(m));
                REP8(X) REP8(X)
#undef X
            } else if (OP == 52) {  // clustered in one asm statement: 8 cndmask then 8 fma, twice
                asm volatile("v_cndmask_b32_e64 %0, %0, %8, %10\n v_cndmask_b32_e64 %1, %1, %8, %10\n v_cndmask_b32_e64 %2, %2, %8, %10\n v_cndmask_b32_e64 %3, %3, %8, %10\n"
                             "v_cndmask_b32_e64 %4, %4, %8, %10\n v_cndmask_b32_e64 %5, %5, %8, %10\n v_cndmask_b32_e64 %6, %6, %8, %10\n v_cndmask_b32_e64 %7, %7, %8, %10\n"
                             "v_cndmask_b32_e64 %0, %0, %8, %10\n v_cndmask_b32_e64 %1, %1, %8, %10\n v_cndmask_b32_e64 %2, %2, %8, %10\n v_cndmask_b32_e64 %3, %3, %8, %10\n"
                             "v_cndmask_b32_e64 %4, %4, %8, %10\n v_cndmask_b32_e64 %5, %5, %8, %10\n v_cndmask_b32_e64 %6, %6, %8, %10\n v_cndmask_b32_e64 %7, %7, %8, %10\n"
                             "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(m));
            } else if (OP == 53) {  // dependent chain like push2_k3: cmp -> cnd -> cmp -> cnd (2 elems), no fma
                asm volatile("v_cmp_nge_f32_e64 %4, %2, %0\n v_cmp_nge_f32_e64 %5, %2, %1\n s_nop 0\n v_cndmask_b32_e64 %6, %2, %3, %4\n v_cndmask_b32_e64 %7, %2, %3, %5\n"
                             "v_cmp_nge_f32_e64 %4, %6, %0\n v_cmp_nge_f32_e64 %5, %7, %1\n s_nop 0\n v_cndmask_b32_e64 %6, %2, %3, %4\n v_cndmask_b32_e64 %7, %2, %3, %5\n"
                             "v_cmp_nge_f32_e64 %4, %6, %0\n v_cmp_nge_f32_e64 %5, %7, %1\n s_nop 0\n v_cndmask_b32_e64 %6, %2, %3, %4\n v_cndmask_b32_e64 %7, %2, %3, %5\n"
                             "v_cmp_nge_f32_e64 %4, %6, %0\n v_cmp_nge_f32_e64 %5, %7, %1\n s_nop 0\n v_cndmask_b32_e64 %0, %2, %3, %4\n v_cndmask_b32_e64 %1, %2, %3, %5"
                             : "+v"(a0), "+v"(a1) : "v"(b), "v"(c), "s"(m), "s"(m2), "v"(a2), "v"(a3) : "vcc");
            } else if (OP == 60) {  // v_fma with SGPR operand
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "s"(sc), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 61) {  // v_fmaak (literal K)
#define X(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f9d70a4" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 62) {  // v_fma with SGPR + abs
#define X(i) asm volatile("v_fma_f32 %0, %0, |%2|, %1" : "+v"(a##i) : "s"(sc), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 63) {  // v_fmac (VOP2) with SGPR src0
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##i) : "s"(sc), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 64) {  // v_fma VOP3 with inline constant
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, -1.0" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (OP == 65) {  // v_mul with SGPR
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a##i) : "s"(sc));
                REP8(X)
#undef X
            } else if (OP == 66) {  // v_fmamk
#define X(i) asm volatile("v_fmamk_f32 %0, %0, 0x3f9d70a4, %1" : "+v"(a##i) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 70) {  // pairs: v_cmp->sgpr + fma
#define X(i) asm volatile("v_cmp_nge_f32_e64 %1, %0, %2\n v_fma_f32 %0, %0, %2, %3" : "+v"(a##i), "=s"(m) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 71) {  // pairs: v_addc(sgpr carry) + fma
#define X(i) asm volatile("v_addc_co_u32_e64 %0, vcc, %0, %0, %1\n v_fma_f32 %0, %0, %2, %3" : "+v"(a##i) : "s"(m), "v"(b), "v"(c) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 72) {  // pairs: v_max + fma
#define X(i) asm volatile("v_max_f32 %0, %0, %1\n v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 73) {  // triples: cmp, cndmask, addc then 3 fma
#define X(i) asm volatile("v_cmp_nge_f32_e64 %1, %0, %2\n v_fma_f32 %0, %0, %2, %3\n s_nop 0\n v_cndmask_b32_e64 %0, %0, %2, %1\n v_fma_f32 %0, %0, %2, %3\n v_addc_co_u32_e64 %0, vcc, %0, %0, %1\n v_fma_f32 %0, %0, %2, %3" : "+v"(a##i), "=&s"(m) : "v"(b), "v"(c) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 80 || OP == 81 || OP == 82) {
                unsigned long long mAa, mAb, mBa, mBb, mCa, mCb; float ta, tb, pa, pb, qa, qb; unsigned wacc = 0;
                if (OP == 80) { asm volatile("v_cmp_nge_f32_e64 %[mAa], %[b3], %[xa]\nv_cmp_nge_f32_e64 %[mAb], %[b3], %[xb]\nv_cndmask_b32_e64 %[pa], %[b0], %[b4], %[mAa]\nv_cndmask_b32_e64 %[pb], %[b0], %[b4], %[mAb]\nv_cndmask_b32_e64 %[ta], %[b1], %[b5], %[mAa]\nv_cndmask_b32_e64 %[tb], %[b1], %[b5], %[mAb]\nv_cmp_nge_f32_e64 %[mBa], %[ta], %[xa]\nv_cmp_nge_f32_e64 %[mBb], %[tb], %[xb]\nv_cndmask_b32_e64 %[qa], %[b2], %[b6], %[mAa]\nv_cndmask_b32_e64 %[qb], %[b2], %[b6], %[mAb]\nv_cndmask_b32_e64 %[ta], %[pa], %[qa], %[mBa]\nv_cndmask_b32_e64 %[tb], %[pb], %[qb], %[mBb]\nv_cmp_nge_f32_e64 %[mCa], %[ta], %[xa]\nv_cmp_nge_f32_e64 %[mCb], %[tb], %[xb]\nv_cndmask_b32_e64 %[pa], 0, 4, %[mAa]\nv_cndmask_b32_e64 %[pb], 0, 4, %[mAb]\nv_cndmask_b32_e64 %[qa], 0, 2, %[mBa]\nv_cndmask_b32_e64 %[qb], 0, 2, %[mBb]\nv_cndmask_b32_e64 %[ta], 0, 1, %[mCa]\nv_cndmask_b32_e64 %[tb], 0, 1, %[mCb]\nv_or3_b32 %[pa], %[pa], %[qa], %[ta]\nv_or3_b32 %[pb], %[pb], %[qb], %[tb]\nv_lshl_or_b32 %[w], %[w], 3, %[pa]\nv_lshl_or_b32 %[w], %[w], 3, %[pb]" : [w] "+v"(wacc), [f0] "+v"(a4), [f1] "+v"(a5), [f2] "+v"(a6), [f3] "+v"(a7), [mAa] "=&s"(mAa), [mAb] "=&s"(mAb), [mBa] "=&s"(mBa), [mBb] "=&s"(mBb), [mCa] "=&s"(mCa), [mCb] "=&s"(mCb), [ta] "=&v"(ta), [tb] "=&v"(tb), [pa] "=&v"(pa), [pb] "=&v"(pb), [qa] "=&v"(qa), [qb] "=&v"(qb) : [xa] "v"(a0), [xb] "v"(a1), [b0] "v"(b), [b1] "v"(c), [b2] "v"(a2), [b3] "v"(a3), [b4] "v"(b), [b5] "v"(c), [b6] "v"(a2)); }
                if (OP == 81) { asm volatile("v_cmp_nge_f32_e64 %[mAa], %[b3], %[xa]\nv_fma_f32 %[f0], %[f0], %[xa], %[xb]\nv_cmp_nge_f32_e64 %[mAb], %[b3], %[xb]\nv_fma_f32 %[f1], %[f1], %[xa], %[xb]\nv_cndmask_b32_e64 %[pa], %[b0], %[b4], %[mAa]\nv_fma_f32 %[f2], %[f2], %[xa], %[xb]\nv_cndmask_b32_e64 %[pb], %[b0], %[b4], %[mAb]\nv_fma_f32 %[f3], %[f3], %[xa], %[xb]\nv_cndmask_b32_e64 %[ta], %[b1], %[b5], %[mAa]\nv_fma_f32 %[f0], %[f0], %[xa], %[xb]\nv_cndmask_b32_e64 %[tb], %[b1], %[b5], %[mAb]\nv_fma_f32 %[f1], %[f1], %[xa], %[xb]\nv_cmp_nge_f32_e64 %[mBa], %[ta], %[xa]\nv_fma_f32 %[f2], %[f2], %[xa], %[xb]\nv_cmp_nge_f32_e64 %[mBb], %[tb], %[xb]\nv_fma_f32 %[f3], %[f3], %[xa], %[xb]\nv_cndmask_b32_e64 %[qa], %[b2], %[b6], %[mAa]\nv_fma_f32 %[f0], %[f0], %[xa], %[xb]\nv_cndmask_b32_e64 %[qb], %[b2], %[b6], %[mAb]\nv_fma_f32 %[f1], %[f1], %[xa], %[xb]\nv_cndmask_b32_e64 %[ta], %[pa], %[qa], %[mBa]\nv_fma_f32 %[f2], %[f2], %[xa], %[xb]\nv_cndmask_b32_e64 %[tb], %[pb], %[qb], %[mBb]\nv_fma_f32 %[f3], %[f3], %[xa], %[xb]\nv_cmp_nge_f32_e64 %[mCa], %[ta], %[xa]\nv_fma_f32 %[f0], %[f0], %[xa], %[xb]\nv_cmp_nge_f32_e64 %[mCb], %[tb], %[xb]\nv_fma_f32 %[f1], %[f1], %[xa], %[xb]\nv_cndmask_b32_e64 %[pa], 0, 4, %[mAa]\nv_fma_f32 %[f2], %[f2], %[xa], %[xb]\nv_cndmask_b32_e64 %[pb], 0, 4, %[mAb]\nv_fma_f32 %[f3], %[f3], %[xa], %[xb]\nv_cndmask_b32_e64 %[qa], 0, 2, %[mBa]\nv_fma_f32 %[f0], %[f0], %[xa], %[xb]\nv_cndmask_b32_e64 %[qb], 0, 2, %[mBb]\nv_fma_f32 %[f1], %[f1], %[xa], %[xb]\nv_cndmask_b32_e64 %[ta], 0, 1, %[mCa]\nv_fma_f32 %[f2], %[f2], %[xa], %[xb]\nv_cndmask_b32_e64 %[tb], 0, 1, %[mCb]\nv_fma_f32 %[f3], %[f3], %[xa], %[xb]\nv_or3_b32 %[pa], %[pa], %[qa], %[ta]\nv_fma_f32 %[f0], %[f0], %[xa], %[xb]\nv_or3_b32 %[pb], %[pb], %[qb], %[tb]\nv_fma_f32 %[f1], %[f1], %[xa], %[xb]\nv_lshl_or_b32 %[w], %[w], 3, %[pa]\nv_fma_f32 %[f2], %[f2], %[xa], %[xb]\nv_lshl_or_b32 %[w], %[w], 3, %[pb]\nv_fma_f32 %[f3], %[f3], %[xa], %[xb]" : [w] "+v"(wacc), [f0] "+v"(a4), [f1] "+v"(a5), [f2] "+v"(a6), [f3] "+v"(a7), [mAa] "=&s"(mAa), [mAb] "=&s"(mAb), [mBa] "=&s"(mBa), [mBb] "=&s"(mBb), [mCa] "=&s"(mCa), [mCb] "=&s"(mCb), [ta] "=&v"(ta), [tb] "=&v"(tb), [pa] "=&v"(pa), [pb] "=&v"(pb), [qa] "=&v"(qa), [qb] "=&v"(qb) : [xa] "v"(a0), [xb] "v"(a1), [b0] "v"(b), [b1] "v"(c), [b2] "v"(a2), [b3] "v"(a3), [b4] "v"(b), [b5] "v"(c), [b6] "v"(a2)); }
                if (OP == 82) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
                    REP8(X) REP8(X) REP8(X)
#undef X
                }
                a0 += __builtin_bit_cast(float, wacc & 1u);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(m & 1);
}

template <int OP> int run(const char* name, float* out, int lanes_mult) {
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double winstr = (double)blocks * 4 * iters * 64;  // wave-instructions
    double per_cu_per_ns = winstr / 256 / (ms * 1e6);
    printf("%-18s %8.3f ms  wave-instr/ns/CU = %.3f  (lane-ops/s = %.2f T)\n", name, ms, per_cu_per_ns, winstr * 64 * lanes_mult / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    run<0>("v_fma_f32", out, 1); run<1>("v_pk_fma_f32", out, 2); run<13>("v_mul_f32", out, 1); run<2>("v_exp_f32", out, 1); run<7>("v_rcp_f32", out, 1);
    run<3>("v_cmp_nge->sgpr", out, 1); run<4>("v_cndmask sgpr", out, 1); run<5>("v_addc sgpr", out, 1);
    run<6>("v_cvt_pk_bf16", out, 1); run<8>("v_lshl_or", out, 1); run<9>("v_max_f32", out, 1); run<10>("v_bfe_u32", out, 1);
    run<11>("s_or_b64", out, 1); run<12>("fma+s_or", out, 1);
    run<14>("v_add_f32", out, 1); run<15>("v_pk_add_u16", out, 2); run<16>("v_lshlrev_b32", out, 1); run<17>("v_cvt_f32_f16", out, 1);
    run<80>("bucket S-ops only (24 S)", out, 24); run<81>("bucket 24 S + 24 fma interleaved", out, 48); run<82>("24 fma only", out, 24);
    run<70>("pairs cmp+fma", out, 2); run<71>("pairs addc+fma", out, 2); run<72>("pairs max+fma", out, 2); run<73>("cmp,f,cnd,f,addc,f (6 valu)", out, 6);
    run<60>("v_fma sgpr operand", out, 1); run<61>("v_fmaak literal", out, 1); run<62>("v_fma sgpr + abs", out, 1); run<63>("v_fmac sgpr", out, 1); run<64>("v_fma inline const", out, 1); run<65>("v_mul sgpr", out, 1); run<66>("v_fmamk literal", out, 1);
    run<50>("clustered 16cnd+16fma (x32)", out, 4); run<51>("alternating cnd,fma (x32)", out, 4); run<52>("clustered one-asm (x32)", out, 4); run<53>("dep chain cmp/cnd (x16 valu)", out, 2);
    run<30>("mix fma+cndmask (pairs)", out, 1); run<31>("mix exp+fma (pairs)", out, 1); run<32>("mix exp+cndmask (pairs)", out, 1); run<33>("mix exp+3fma (per 4)", out, 1);
    run<34>("v_or_b32", out, 1); run<35>("v_xor_b32", out, 1); run<36>("v_add_u32", out, 1); run<37>("v_lshl_add_u32", out, 1); run<38>("v_mad_u32_u24", out, 1);
    run<39>("v_sub_f32 clamp", out, 1); run<40>("v_mov_b32", out, 1); run<41>("v_bfi_b32", out, 1); run<42>("v_min_f32", out, 1); run<43>("v_cvt_f32_i32", out, 1);
    run<44>("v_med3_f32", out, 1); run<45>("v_and_or_b32", out, 1); run<46>("v_fma clamp", out, 1); run<47>("v_alignbit", out, 1);
    run<18>("v_and_b32", out, 1); run<19>("v_fma -|a|", out, 1); run<20>("v_fmac_f32", out, 1); run<21>("v_perm_b32", out, 1); run<22>("v_mov dpp quad", out, 1); run<23>("v_cmp -> vcc", out, 1);
    return 0;
}
