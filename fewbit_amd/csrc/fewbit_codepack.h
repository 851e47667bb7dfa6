// fewbit_codepack.h -- "bucket 8 elements against the border table and pack the k-bit codes",
// hand-scheduled for gfx950.
//
// code = #{ j : b_j < x } for sorted borders == MSB-first binary search with predicate !(b >= x)
// (true for NaN, so NaN -> 2^K-1: torch.searchsorted's CPU rule, fewbit/cpu/gelu.cc:17 in the
// reference).  Per element this is K v_cmp + (2^K-1-K) v_cndmask to walk the tree and K v_addc to
// shift the decided bits into the packed word (w = 2w + bit, elements 7..0, MSB first), i.e.
// 2^K-1+K VALU instructions, none of them dependent on a table in memory: the borders are
// wave-uniform values held in VGPRs (v_cndmask takes one scalar operand, the lane mask).
//
// Why inline asm: hipcc turns `w = 2*w + bit` into v_cndmask + v_lshl_or (2 instructions per
// bit) and duplicates negated compares; measured on MI355X these "slow class" VALU instructions
// (v_cmp / v_cndmask / v_addc, ~4.1 cycles per wave64 each against ~2.6 for v_fma) are the largest
// single cost of the forward kernel, so their count is worth pinning.
//
// Hazard (gfx940+, LLVM "VALUWriteSGPRVALURead"): a VALU instruction that reads an SGPR/VCC written
// by a VALU instruction needs 2 wait states in between.  hipcc does not pad inside an asm
// statement, so every block below is scheduled by hand: elements are interleaved so that at least
// two instructions separate each v_cmp from the first v_cndmask / v_addc consuming its mask, and
// an s_nop fills in where no independent instruction is available.  Each statement only reads its
// inputs and writes its own outputs/temporaries (early-clobber), so it is not `volatile`.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fewbit_hip {

typedef unsigned long long lanemask_t;

// ---- K = 1: w = 2w + !(b0 >= x), four elements per statement --------------------------------
__device__ __forceinline__ uint32_t push4_k1(uint32_t w, float x3, float x2, float x1, float x0, float b0) {
    lanemask_t m3, m2, m1, m0;
    asm("v_cmp_nge_f32_e64 %[m3], %[b0], %[x3]\n\t"
        "v_cmp_nge_f32_e64 %[m2], %[b0], %[x2]\n\t"
        "v_cmp_nge_f32_e64 %[m1], %[b0], %[x1]\n\t"
        "v_cmp_nge_f32_e64 %[m0], %[b0], %[x0]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[m3]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[m2]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[m1]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[m0]"
        : [w] "+v"(w), [m3] "=&s"(m3), [m2] "=&s"(m2), [m1] "=&s"(m1), [m0] "=&s"(m0)
        : [x3] "v"(x3), [x2] "v"(x2), [x1] "v"(x1), [x0] "v"(x0), [b0] "v"(b0)
        : "vcc");
    return w;
}

// ---- K = 2: borders b0 < b1 < b2; two elements (xa first = more significant) -----------------
__device__ __forceinline__ uint32_t push2_k2(uint32_t w, float xa, float xb, const float (&b)[3]) {
    lanemask_t mAa, mAb, mBa, mBb;
    float ta, tb;
    asm("v_cmp_nge_f32_e64 %[mAa], %[b1], %[xa]\n\t"
        "v_cmp_nge_f32_e64 %[mAb], %[b1], %[xb]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_e64 %[ta], %[b0], %[b2], %[mAa]\n\t"
        "v_cndmask_b32_e64 %[tb], %[b0], %[b2], %[mAb]\n\t"
        "v_cmp_nge_f32_e64 %[mBa], %[ta], %[xa]\n\t"
        "v_cmp_nge_f32_e64 %[mBb], %[tb], %[xb]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mAa]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mBa]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mAb]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mBb]"
        : [w] "+v"(w), [mAa] "=&s"(mAa), [mAb] "=&s"(mAb), [mBa] "=&s"(mBa), [mBb] "=&s"(mBb), [ta] "=&v"(ta),
          [tb] "=&v"(tb)
        : [xa] "v"(xa), [xb] "v"(xb), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2])
        : "vcc");
    return w;
}

// ---- K = 3: borders b0..b6; two elements per statement, 20 VALU + 1 s_nop --------------------
//   A = !(b3 >= x);  t = A ? b5 : b1;  B = !(t >= x);
//   p = B ? b2 : b0; q = B ? b6 : b4;  t = A ? q : p;  C = !(t >= x)
__device__ __forceinline__ uint32_t push2_k3(uint32_t w, float xa, float xb, const float (&b)[7]) {
    lanemask_t mAa, mAb, mBa, mBb, mCa, mCb;
    float ta, tb, pa, qa, pb, qb;
    asm("v_cmp_nge_f32_e64 %[mAa], %[b3], %[xa]\n\t"
        "v_cmp_nge_f32_e64 %[mAb], %[b3], %[xb]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_e64 %[ta], %[b1], %[b5], %[mAa]\n\t"
        "v_cndmask_b32_e64 %[tb], %[b1], %[b5], %[mAb]\n\t"
        "v_cmp_nge_f32_e64 %[mBa], %[ta], %[xa]\n\t"
        "v_cmp_nge_f32_e64 %[mBb], %[tb], %[xb]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mAa]\n\t"
        "v_cndmask_b32_e64 %[pa], %[b0], %[b2], %[mBa]\n\t"
        "v_cndmask_b32_e64 %[qa], %[b4], %[b6], %[mBa]\n\t"
        "v_cndmask_b32_e64 %[pb], %[b0], %[b2], %[mBb]\n\t"
        "v_cndmask_b32_e64 %[qb], %[b4], %[b6], %[mBb]\n\t"
        "v_cndmask_b32_e64 %[ta], %[pa], %[qa], %[mAa]\n\t"
        "v_cmp_nge_f32_e64 %[mCa], %[ta], %[xa]\n\t"
        "v_cndmask_b32_e64 %[tb], %[pb], %[qb], %[mAb]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mBa]\n\t"
        "v_cmp_nge_f32_e64 %[mCb], %[tb], %[xb]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mCa]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mAb]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mBb]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mCb]"
        : [w] "+v"(w), [mAa] "=&s"(mAa), [mAb] "=&s"(mAb), [mBa] "=&s"(mBa), [mBb] "=&s"(mBb), [mCa] "=&s"(mCa),
          [mCb] "=&s"(mCb), [ta] "=&v"(ta), [tb] "=&v"(tb), [pa] "=&v"(pa), [qa] "=&v"(qa), [pb] "=&v"(pb),
          [qb] "=&v"(qb)
        : [xa] "v"(xa), [xb] "v"(xb), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]),
          [b5] "v"(b[5]), [b6] "v"(b[6])
        : "vcc");
    return w;
}

// ---- K = 4: borders b0..b14; one element per statement, 19 VALU + 2 s_nop --------------------
//   A = !(b7 >= x)
//   t = A ? b11 : b3;                       B = !(t >= x)
//   p = A ? b9 : b1;  q = A ? b13 : b5;     t = B ? q : p;        C = !(t >= x)
//   r0..r3 = A ? (b8,b10,b12,b14) : (b0,b2,b4,b6);  r0 = B ? r2 : r0;  r1 = B ? r3 : r1;
//   t = C ? r1 : r0;                        D = !(t >= x)
__device__ __forceinline__ uint32_t push1_k4(uint32_t w, float x, const float (&b)[15]) {
    lanemask_t mA, mB, mC, mD;
    float t, p, q, r0, r1, r2, r3;
    asm("v_cmp_nge_f32_e64 %[mA], %[b7], %[x]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %[t], %[b3], %[b11], %[mA]\n\t"
        "v_cndmask_b32_e64 %[p], %[b1], %[b9], %[mA]\n\t"
        "v_cndmask_b32_e64 %[q], %[b5], %[b13], %[mA]\n\t"
        "v_cmp_nge_f32_e64 %[mB], %[t], %[x]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mA]\n\t"
        "v_cndmask_b32_e64 %[r0], %[b0], %[b8], %[mA]\n\t"
        "v_cndmask_b32_e64 %[r1], %[b2], %[b10], %[mA]\n\t"
        "v_cndmask_b32_e64 %[r2], %[b4], %[b12], %[mA]\n\t"
        "v_cndmask_b32_e64 %[r3], %[b6], %[b14], %[mA]\n\t"
        "v_cndmask_b32_e64 %[t], %[p], %[q], %[mB]\n\t"
        "v_cmp_nge_f32_e64 %[mC], %[t], %[x]\n\t"
        "v_cndmask_b32_e64 %[r0], %[r0], %[r2], %[mB]\n\t"
        "v_cndmask_b32_e64 %[r1], %[r1], %[r3], %[mB]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mB]\n\t"
        "v_cndmask_b32_e64 %[t], %[r0], %[r1], %[mC]\n\t"
        "v_cmp_nge_f32_e64 %[mD], %[t], %[x]\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mC]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %[w], vcc, %[w], %[w], %[mD]"
        : [w] "+v"(w), [mA] "=&s"(mA), [mB] "=&s"(mB), [mC] "=&s"(mC), [mD] "=&s"(mD), [t] "=&v"(t), [p] "=&v"(p),
          [q] "=&v"(q), [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3)
        : [x] "v"(x), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]),
          [b6] "v"(b[6]), [b7] "v"(b[7]), [b8] "v"(b[8]), [b9] "v"(b[9]), [b10] "v"(b[10]), [b11] "v"(b[11]),
          [b12] "v"(b[12]), [b13] "v"(b[13]), [b14] "v"(b[14])
        : "vcc");
    return w;
}

// ---- plain C++ formulation (build with -DFEWBIT_CXX_BUCKET; kept as the readable statement of the algorithm and
// as an A/B): the same MSB-first search, bits materialised with selects of inline constants (v_cndmask 0/4, 0/2,
// 0/1 + v_or3 + v_lshl_or) instead of v_addc.  In isolation it is ~5 % faster than the blocks above (v_addc with an
// SGPR carry-in does not co-issue with v_fma, v_cmp/v_cndmask do), but inside the forward kernel hipcc's schedule
// of it needs more than 64 VGPRs and spills (4096x4096 bf16 forward 16.2 -> 18.0 us), so the asm blocks are used.
template <int K> __device__ __forceinline__ uint32_t bucket(const float (&b)[(1 << K) - 1], float x) {
    if constexpr (K == 1) {
        return !(b[0] >= x) ? 1u : 0u;
    } else if constexpr (K == 2) {
        const bool A = !(b[1] >= x);
        const float t = A ? b[2] : b[0];
        const bool B = !(t >= x);
        return (A ? 2u : 0u) | (B ? 1u : 0u);
    } else if constexpr (K == 3) {
        const bool A = !(b[3] >= x);
        const float t = A ? b[5] : b[1];
        const bool B = !(t >= x);
        const float p = A ? b[4] : b[0];
        const float q = A ? b[6] : b[2];
        const float t0 = B ? q : p;
        const bool C = !(t0 >= x);
        return (A ? 4u : 0u) | (B ? 2u : 0u) | (C ? 1u : 0u);
    } else {
        const bool A = !(b[7] >= x);
        const float t = A ? b[11] : b[3];
        const bool B = !(t >= x);
        const float p = A ? b[9] : b[1];
        const float q = A ? b[13] : b[5];
        const float t2 = B ? q : p;
        const bool C = !(t2 >= x);
        const float r0 = A ? b[8] : b[0], r1 = A ? b[10] : b[2], r2 = A ? b[12] : b[4], r3 = A ? b[14] : b[6];
        const float s0 = B ? r2 : r0, s1 = B ? r3 : r1;
        const float t3 = C ? s1 : s0;
        const bool D = !(t3 >= x);
        return (A ? 8u : 0u) | (B ? 4u : 0u) | (C ? 2u : 0u) | (D ? 1u : 0u);
    }
}

// packed K-bit codes of one group (8 elements), element i in bits [K*i, K*i+K)
template <int K> __device__ __forceinline__ uint32_t pack_group(const float (&x)[8], const float (&b)[(1 << K) - 1]) {
    uint32_t w = 0;
#if !defined(FEWBIT_CXX_BUCKET)
    if constexpr (K == 1) {
        w = push4_k1(w, x[7], x[6], x[5], x[4], b[0]);
        w = push4_k1(w, x[3], x[2], x[1], x[0], b[0]);
    } else if constexpr (K == 2) {
#pragma unroll
        for (int i = 7; i > 0; i -= 2) w = push2_k2(w, x[i], x[i - 1], b);
    } else if constexpr (K == 3) {
#pragma unroll
        for (int i = 7; i > 0; i -= 2) w = push2_k3(w, x[i], x[i - 1], b);
    } else {
#pragma unroll
        for (int i = 7; i >= 0; --i) w = push1_k4(w, x[i], b);
    }
#else
#pragma unroll
    for (int i = 0; i < 8; ++i) w |= bucket<K>(b, x[i]) << (K * i);
#endif
    return w;
}

// packed K-bit codes of HALF a group (4 elements), element i in bits [K*i, K*i+K): the fp32 split layout packs the two
// halves of a group in two neighbouring lanes (fewbit_device.h, SplitF32)
template <int K> __device__ __forceinline__ uint32_t pack_half(const float (&x)[4], const float (&b)[(1 << K) - 1]) {
    uint32_t w = 0;
#if !defined(FEWBIT_CXX_BUCKET)
    if constexpr (K == 1) {
        w = push4_k1(w, x[3], x[2], x[1], x[0], b[0]);
    } else if constexpr (K == 2) {
        w = push2_k2(w, x[3], x[2], b);
        w = push2_k2(w, x[1], x[0], b);
    } else if constexpr (K == 3) {
        w = push2_k3(w, x[3], x[2], b);
        w = push2_k3(w, x[1], x[0], b);
    } else {
#pragma unroll
        for (int i = 3; i >= 0; --i) w = push1_k4(w, x[i], b);
    }
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) w |= bucket<K>(b, x[i]) << (K * i);
#endif
    return w;
}

}  // namespace fewbit_hip
