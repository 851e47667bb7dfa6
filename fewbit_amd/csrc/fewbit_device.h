// fewbit_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// Written for CDNA4 only (wave64, v_cvt_pk_bf16_f32, DPP); there is no other target.
// Vocabulary: a GROUP is 8 consecutive elements; its k-bit codes occupy exactly k bytes of the
// packed state (fewbit/cpu/codec.h:33-57 in the reference), so groups are the unit of work.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fewbit_hip.h"

namespace fewbit_hip {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
// occupancy the streaming kernels are built for: 8 waves per SIMD = 32 per CU, i.e. at most 64 VGPRs.
// (One VGPR over and the hardware admits 7 blocks per CU; the 8th then runs as a second round that
// nearly doubles the duration of a ~10 us kernel -- seen with per-wave timestamps.)
constexpr int kWavesPerSimd = 8;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// Product rounded to fp32 and kept as such.  Without the barrier hipcc contracts (fp16 -> fp32) * m -> fp16 into
// v_fma_mixlo_f16 with a +0 addend, which turns a -0 product into +0 (the gradient of a negative gy through a
// zero level must stay -0 to match the reference bit for bit).
__device__ __forceinline__ float mul_f32(float a, float b) {
    float r = a * b;
    asm("" : "+v"(r));
    return r;
}

// ------------------------------------------------------------------ scalar element I/O
template <int DT> struct Elem;

template <> struct Elem<FEWBIT_F32> {
    typedef float type;
    static constexpr int kSize = 4;
    static __device__ __forceinline__ float load(const void *p, size_t i) { return static_cast<const float *>(p)[i]; }
    static __device__ __forceinline__ void store(void *p, size_t i, float v) { static_cast<float *>(p)[i] = v; }
};

template <> struct Elem<FEWBIT_F16> {
    typedef _Float16 type;
    static constexpr int kSize = 2;
    static __device__ __forceinline__ float load(const void *p, size_t i) {
        return static_cast<float>(static_cast<const _Float16 *>(p)[i]);
    }
    static __device__ __forceinline__ void store(void *p, size_t i, float v) {
        static_cast<_Float16 *>(p)[i] = static_cast<_Float16>(v);  // v_cvt_f16_f32, RNE
    }
};

template <> struct Elem<FEWBIT_BF16> {
    typedef uint16_t type;
    static constexpr int kSize = 2;
    static __device__ __forceinline__ float load(const void *p, size_t i) {
        return bits_f32(static_cast<uint32_t>(static_cast<const uint16_t *>(p)[i]) << 16);
    }
    static __device__ __forceinline__ void store(void *p, size_t i, float v) {
        static_cast<__bf16 *>(p)[i] = static_cast<__bf16>(v);  // v_cvt_pk_bf16_f32, RNE
    }
};

// ------------------------------------------------------------------ whole-group (8 element) I/O
// 16-bit dtypes: one 16-byte access per lane, lane-contiguous -> each wave instruction moves 1 KiB
// of contiguous memory.  fp32: two 16-byte accesses per lane (32-byte lane stride).
// `Raw` is the group as loaded (4 or 8 VGPRs): the software pipeline prefetches the next tile in this
// form and only unpacks to 8 floats when the tile is computed.
template <int DT> struct GroupIO;

// Stores take an NT flag.  What the kernels use (all measured on MI355X):
//   y, gx of 16-bit dtypes: nontemporal -- not read again by this path; write-allocated they push the still-to-be-read
//                           input out of L2 / Infinity Cache and are written back during the next kernel (forward
//                           -1.2 us at 4096x4096; backward cache-cold 15.3 -> 13.9 us, 2^26 elements step 100.6 -> 94.1 us)
//   fp32 y, gx            : plain -- fp32 groups are two 16 B pieces at a 32 B lane stride, holes that only L2
//                           write-combining fills (nontemporal: +4 us)
//   packed state          : plain -- it is what backward reads next (-0.9 us per step)
// Loads are always plain (nontemporal loads measured +2 us).
//
// Alignment: gfx950 global memory instructions take any byte address (hipcc itself emits global_load_dwordx4 for
// an align-1 vector), so every access below goes through a typedef that only promises the element's own alignment
// (fp32 data: 4, 16-bit data: 2, state: 1).  The instructions are the same as for 16-byte aligned pointers, and a
// tensor view that starts at an odd element offset takes the same kernels at the same speed.
template <typename T, int A> struct Unaligned { typedef T type __attribute__((aligned(A))); };
template <int A, typename T> __device__ __forceinline__ T load_as(const void *p) {
    return *static_cast<const typename Unaligned<T, A>::type *>(p);
}
template <bool NT, int A, typename T> __device__ __forceinline__ void store_as(void *p, T v) {
    typename Unaligned<T, A>::type *q = static_cast<typename Unaligned<T, A>::type *>(p);
    if constexpr (NT) __builtin_nontemporal_store(v, q);
    else *q = v;
}

template <> struct GroupIO<FEWBIT_F32> {
    struct Raw { f32x4 a, b; };
    static __device__ __forceinline__ Raw load_raw(const void *base, size_t g) {
        const float *p = static_cast<const float *>(base) + 8 * g;
        return Raw{load_as<4, f32x4>(p), load_as<4, f32x4>(p + 4)};
    }
    static __device__ __forceinline__ void unpack(const Raw &r, float (&v)[8]) {
        v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w;
        v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
    }
    template <bool NT = false>
    static __device__ __forceinline__ void store(void *base, size_t g, const float (&v)[8]) {
        float *p = static_cast<float *>(base) + 8 * g;
        f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        store_as<NT, 4>(p, a);
        store_as<NT, 4>(p + 4, b);
    }
};

template <> struct GroupIO<FEWBIT_BF16> {
    struct Raw { u32x4 q; };
    static __device__ __forceinline__ Raw load_raw(const void *base, size_t g) {
        return Raw{load_as<2, u32x4>(static_cast<const uint16_t *>(base) + 8 * g)};
    }
    static __device__ __forceinline__ void unpack(const Raw &r, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = bits_f32(r.q[i] << 16);
            v[2 * i + 1] = bits_f32(r.q[i] & 0xffff0000u);
        }
    }
    template <bool NT = false>
    static __device__ __forceinline__ void store(void *base, size_t g, const float (&v)[8]) {
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x2 f = {v[2 * i], v[2 * i + 1]};
            w[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));  // v_cvt_pk_bf16_f32
        }
        store_as<NT, 2>(static_cast<uint16_t *>(base) + 8 * g, w);
    }
};

template <> struct GroupIO<FEWBIT_F16> {
    struct Raw { u32x4 q; };
    static __device__ __forceinline__ Raw load_raw(const void *base, size_t g) {
        return Raw{load_as<2, u32x4>(static_cast<const uint16_t *>(base) + 8 * g)};
    }
    static __device__ __forceinline__ void unpack(const Raw &r, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(r.q[i] & 0xffffu)));
            v[2 * i + 1] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(r.q[i] >> 16)));
        }
    }
    template <bool NT = false>
    static __device__ __forceinline__ void store(void *base, size_t g, const float (&v)[8]) {
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x2 f = {v[2 * i], v[2 * i + 1]};
            w[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2));  // v_cvt_pk_f16_f32, RNE
        }
        store_as<NT, 2>(static_cast<uint16_t *>(base) + 8 * g, w);
    }
};

// ------------------------------------------------------------------ fp32 tiles in the SPLIT layout
// GroupIO<FEWBIT_F32> gives lane l the 32 contiguous bytes of group l as two 16 B accesses at a 32 B lane stride: every
// wave instruction touches 2 KiB and uses every other 16 B piece of it -- twice the cache lines per instruction, and
// stores that only L2 write-combining turns into full lines (so no nontemporal stores).  In the split layout a tile of 64
// groups (2 KiB) is moved by two fully contiguous 1 KiB instructions: lane l takes 16 B piece l and piece 64 + l, i.e.
// HALF (l & 1) of group l/2 and the same half of group 32 + l/2.  Values need no exchange at all (they are stored back
// the way they came); only the 4K-bit half-words of codes cross lanes: one DPP move per half pairs them up with the
// neighbouring lane's, and one ds_bpermute puts the word of group t into lane t, where the quad state I/O expects it
// (and takes it from there in the backward).
#ifndef FEWBIT_F32_SPLIT
#define FEWBIT_F32_SPLIT 1
#endif
struct SplitF32 {
    struct Raw { f32x4 a, b; };
    // g = first group of the tile + ln (ln = lane, or 0 for the collapsed redirect of pipeline2)
    static __device__ __forceinline__ Raw load_raw(const void *base, size_t g, int ln) {
        const float *p = static_cast<const float *>(base) + 8 * (g - ln) + 4 * ln;
        return Raw{load_as<4, f32x4>(p), load_as<4, f32x4>(p + 256)};
    }
    static __device__ __forceinline__ void unpack(const Raw &r, float (&v)[8]) {
        v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w;      // half (lane & 1) of group  lane/2
        v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;      // the same half of group 32 + lane/2
    }
    template <bool NT>
    static __device__ __forceinline__ void store(void *base, size_t g, int lane, const float (&v)[8]) {
        float *p = static_cast<float *>(base) + 8 * (g - lane) + 4 * lane;
        f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        store_as<NT, 4>(p, a);
        store_as<NT, 4>(p + 256, b);
    }
};

// ------------------------------------------------------------------ packed state access, one group
// K bytes at byte offset K*g.
template <int K> __device__ __forceinline__ void store_state(uint8_t *state, size_t g, uint32_t w) {
    uint8_t *p = state + static_cast<size_t>(K) * g;
    if constexpr (K == 1) {
        p[0] = static_cast<uint8_t>(w);
    } else if constexpr (K == 2) {
        store_as<false, 1>(p, static_cast<uint16_t>(w));
    } else if constexpr (K == 3) {
        store_as<false, 1>(p, static_cast<uint16_t>(w));
        p[2] = static_cast<uint8_t>(w >> 16);
    } else {
        store_as<false, 1>(p, w);
    }
}

template <int K> __device__ __forceinline__ uint32_t load_state(const uint8_t *state, size_t g) {
    const uint8_t *p = state + static_cast<size_t>(K) * g;
    if constexpr (K == 1) {
        return p[0];
    } else if constexpr (K == 2) {
        return load_as<1, uint16_t>(p);
    } else if constexpr (K == 3) {
        return static_cast<uint32_t>(load_as<1, uint16_t>(p)) | (static_cast<uint32_t>(p[2]) << 16);
    } else {
        return load_as<1, uint32_t>(p);
    }
}

// ------------------------------------------------------------------ packed state access, quad of groups
// Fast-path form.  Four adjacent lanes (a quad) own four adjacent groups = 4*K contiguous state bytes =
// K dwords.  The quad exchanges its 8K-bit words with DPP quad_perm moves (VALU cross-lane, no LDS) so
// that lane i ends up holding dword min(i, K-1) of the quad; ALL lanes then store/load a naturally
// aligned dword (lanes beyond K-1 duplicate a neighbour: same address, same value).  No byte or short
// accesses, and no exec-divergent branch around a memory instruction -- the latter matters because a
// VMEM op issued on only some paths makes hipcc's s_waitcnt counting conservative and would serialise
// the software pipeline.
#define FEWBIT_QUAD_PERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

template <int CTRL> __device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), CTRL, 0xf, 0xf, true));
}

// byte offset (relative to the state of group g - (lane&3)) handled by this lane, and the dword it holds
template <int K, bool NT = false>
__device__ __forceinline__ void store_state_quad(uint8_t *state, size_t g, int lane, uint32_t w) {
    const int i = lane & 3;
    uint8_t *quad = state + static_cast<size_t>(K) * (g - i);  // 4*K-byte aligned when state is dword aligned
    if constexpr (K == 1) {
        uint32_t t = quad_perm<FEWBIT_QUAD_PERM(0, 0, 2, 2)>(w) | (quad_perm<FEWBIT_QUAD_PERM(1, 1, 3, 3)>(w) << 8);
        uint32_t d = quad_perm<FEWBIT_QUAD_PERM(0, 0, 0, 0)>(t) | (quad_perm<FEWBIT_QUAD_PERM(2, 2, 2, 2)>(t) << 16);
        store_as<NT, 1>(quad, d);
    } else if constexpr (K == 2) {
        uint32_t d = quad_perm<FEWBIT_QUAD_PERM(0, 0, 2, 2)>(w) | (quad_perm<FEWBIT_QUAD_PERM(1, 1, 3, 3)>(w) << 16);
        store_as<NT, 1>(quad + 4 * (i >> 1), d);
    } else if constexpr (K == 3) {
        // C = w0 | w1<<24 | w2<<48 | w3<<72;  dword j = (w_j >> 8j) | (w_{j+1} << (24-8j)),  j = min(i,2)
        const int j = i < 2 ? i : 2;
        const uint32_t lo = quad_perm<FEWBIT_QUAD_PERM(0, 1, 2, 2)>(w);
        const uint32_t hi = quad_perm<FEWBIT_QUAD_PERM(1, 2, 3, 3)>(w);
        const uint32_t d = (lo >> (8 * j)) | (hi << (24 - 8 * j));
        store_as<NT, 1>(quad + 4 * j, d);
    } else {
        store_as<NT, 1>(state + 4 * g, w);
    }
}

template <int K> __device__ __forceinline__ uint32_t load_state_quad_raw(const uint8_t *state, size_t g, int lane) {
    const int i = lane & 3;
    const uint8_t *quad = state + static_cast<size_t>(K) * (g - i);
    if constexpr (K == 1) return load_as<1, uint32_t>(quad);
    else if constexpr (K == 2) return load_as<1, uint32_t>(quad + 4 * (i >> 1));
    else if constexpr (K == 3) return load_as<1, uint32_t>(quad + 4 * (i < 2 ? i : 2));
    else return load_as<1, uint32_t>(state + 4 * g);
}

// turn the dword fetched by load_state_quad_raw into this lane's 8K-bit word (bits above 8K are junk)
template <int K> __device__ __forceinline__ uint32_t load_state_quad_fix(uint32_t raw, int lane) {
    const int i = lane & 3;
    if constexpr (K == 1) return raw >> (8 * i);
    else if constexpr (K == 2) return raw >> (16 * (i & 1));
    else if constexpr (K == 3) {
        // lane holds dword min(i,2); need bits [24i, 24i+24) of the quad: dwords {0,0,1,2} and the next one
        const uint32_t a = quad_perm<FEWBIT_QUAD_PERM(0, 0, 1, 2)>(raw);
        const uint32_t b = quad_perm<FEWBIT_QUAD_PERM(1, 1, 2, 2)>(raw);
        return __builtin_amdgcn_alignbit(b, a, (24 * i) & 31);
    } else return raw;
}

// ------------------------------------------------------------------ split layout: half-words <-> group words
// forward: this lane's two 4K-bit half-words (cA: its half of group lane/2, cB: its half of group 32 + lane/2) -> the
// 8K-bit word of group `lane`
template <int K> __device__ __forceinline__ uint32_t split_halves_to_word(uint32_t cA, uint32_t cB, int lane) {
    const uint32_t pA = quad_perm<FEWBIT_QUAD_PERM(1, 0, 3, 2)>(cA), pB = quad_perm<FEWBIT_QUAD_PERM(1, 0, 3, 2)>(cB);
    const bool odd = (lane & 1) != 0;
    // even lanes offer the word of group lane/2, odd lanes the word of group 32 + lane/2
    const uint32_t lo = odd ? pB : cA, hi = odd ? cB : pA;
    const uint32_t z = lo | (hi << (4 * K));
    const int src = ((2 * lane) & 63) | (lane >> 5);       // lane t < 32 reads lane 2t, lane t >= 32 reads 2(t-32)+1
    return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src * 4, static_cast<int>(z)));
}

// backward: the word of group `lane` (bits above 8K may be junk) -> this lane's two half-words (junk above 4K bits)
template <int K> __device__ __forceinline__ void split_word_to_halves(uint32_t w, int lane, uint32_t &cA, uint32_t &cB) {
    const uint32_t wA = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((lane >> 1) * 4, static_cast<int>(w)));
    const uint32_t wB = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((32 + (lane >> 1)) * 4, static_cast<int>(w)));
    const int sh = (lane & 1) * 4 * K;
    cA = wA >> sh;
    cB = wB >> sh;
}

// ------------------------------------------------------------------ packed state access, wide codes
// Tables of 5..8 bits: a group's word is 8K <= 64 bits at byte offset K*g, which is not dword aligned in general.
// gfx950 global accesses need no alignment (hipcc emits single dword / dwordx2 instructions for these), so a lane
// reads its word with ONE 8-byte load -- which also fetches up to 3 bytes of the next group: callers keep one group
// of slack before the end of the buffer -- and writes it as a dword plus K-4 single bytes.
__device__ __forceinline__ uint64_t load_state_wide(const uint8_t *state, size_t g, int nbits) {
    return load_as<1, uint64_t>(state + static_cast<size_t>(nbits) * g);
}

__device__ __forceinline__ void store_state_wide(uint8_t *state, size_t g, int nbits, uint64_t w) {
    uint8_t *p = state + static_cast<size_t>(nbits) * g;
    if (nbits < 4) {               // ragged narrow tables (3, 5..7 levels): bytes only
        for (int j = 0; j < nbits; ++j) p[j] = static_cast<uint8_t>(w >> (8 * j));
        return;
    }
    // as few store instructions as the width allows (any address is fine on gfx950): 4 -> dword, 5 -> dword + byte,
    // 6 -> dword + short, 7 -> dword + short + byte, 8 -> one 8-byte store (EXPERIMENTS.md section 3)
    if (nbits == 8) {
        store_as<false, 1>(p, w);
        return;
    }
    store_as<false, 1>(p, static_cast<uint32_t>(w));
    const uint32_t hi = static_cast<uint32_t>(w >> 32);
    if (nbits >= 6) {
        store_as<false, 1>(p + 4, static_cast<uint16_t>(hi));
        if (nbits == 7) p[6] = static_cast<uint8_t>(hi >> 16);
    } else if (nbits == 5) {
        p[4] = static_cast<uint8_t>(hi);
    }
}

// ------------------------------------------------------------------ activation math (fp32)
// Two accuracy classes, chosen by the I/O dtype ("Activation math"; tolerances: section 6):
//   precise : fp32 I/O.  GELU as ATen's x*0.5*(1+erf(x*sqrt(1/2))) with erf_precise (~1 ulp); ocml for the rest.
//   fast    : fp16/bf16 I/O, where the result is rounded to 11/8 significant bits anyway.
//             Branch-free, built from the cheap VALU class (v_fma/v_mul/v_add) plus the hardware
//             transcendentals v_exp_f32 / v_rcp_f32 (1 ulp each).
__device__ __forceinline__ float relu_raw(float x) {
    // v_max_f32 without the canonicalising self-max hipcc puts in front of fmaxf()
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// gelu(x) = relu(x) - |x| * Phi(-|x|),  Phi(-a) = exp2(-1 + a*P(a)),  P of degree 7 fitted by
// tools/fit_gelu.py (max abs error of Phi(-a) in fp32 evaluation 4.5e-8; leading coefficient < 0 so
// the exponent runs to -inf, never +inf, for huge |x|).  11 VALU instructions, one transcendental.
__device__ __forceinline__ float gelu_fast(float x) {
    const float a = __builtin_fabsf(x);
    float r = -2.834918860e-06f;
    r = __builtin_fmaf(r, a, 3.937771180e-05f);
    r = __builtin_fmaf(r, a, -1.861801138e-04f);
    r = __builtin_fmaf(r, a, -1.369371021e-04f);
    r = __builtin_fmaf(r, a, 7.063421421e-03f);
    r = __builtin_fmaf(r, a, -5.249617994e-02f);
    r = __builtin_fmaf(r, a, -4.592081904e-01f);
    r = __builtin_fmaf(r, a, -1.151105165e+00f);
    const float q = __builtin_fmaf(r, a, -1.0f);
    const float h = __builtin_amdgcn_exp2f(q);
    return __builtin_fmaf(-a, h, relu_raw(x));
}

__device__ __forceinline__ float sigmoid_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.44269504088896340736f);
    return __builtin_amdgcn_rcpf(1.0f + e);
}

// erf for the precise class: both pieces evaluated, one select, no branch (ocml's erff branches per lane, which
// serialises a wave whose 64 lanes straddle |z| = 1).  Coefficients from tools/fit_erf.py; error of the fp32
// evaluation <= 1.03 ulp (|z| < T) / 1.13 ulp (|z| >= T) of erf, checked against float64 on the device by
// tests/test_gpu_numerics.py.
//   |z| <  T : erf = z + z*R(z^2)
//   |z| >= T : erf = sign(z) * (1 - exp(-(t + t*Q(t)))),  exp via v_exp_f32 with the rounding error of the
//              product p*log2(e) fed back (hi/lo split), so the large argument does not cost accuracy
__device__ __forceinline__ float erf_precise(float z) {
    const float t = __builtin_fabsf(z);
    const float s = t * t;
    float r = -5.990989157e-04f;
    r = __builtin_fmaf(r, s, 4.993211944e-03f);
    r = __builtin_fmaf(r, s, -2.676664293e-02f);
    r = __builtin_fmaf(r, s, 1.128181741e-01f);
    r = __builtin_fmaf(r, s, -3.761249483e-01f);
    r = __builtin_fmaf(r, s, 1.283791512e-01f);
    const float small = __builtin_fmaf(r, t, t);
    // erf(|z| >= 4) is 1 in fp32; without the clamp the degree-8 product overflows from |z| ~ 3e5 on and the hi/lo split
    // below turns inf - inf into NaN (found by running all 2^32 inputs, scratch/all_fp32_gelu.py).  v_min_f32 returns
    // the other operand for a NaN, which is harmless here: the caller multiplies by x, NaN stays NaN.
    const float tc = __builtin_fminf(t, 6.0f);
    float q = 1.130373221e-05f;
    q = __builtin_fmaf(q, tc, -3.235284530e-04f);
    q = __builtin_fmaf(q, tc, 3.645403776e-03f);
    q = __builtin_fmaf(q, tc, -2.376828715e-02f);
    q = __builtin_fmaf(q, tc, 1.062441021e-01f);
    q = __builtin_fmaf(q, tc, 6.351469755e-01f);
    q = __builtin_fmaf(q, tc, 1.286495626e-01f);
    const float p = __builtin_fmaf(q, tc, tc);
    const float kL = 1.44269502162933349609375f;       // fp32(log2 e)
    const float kLlo = 1.92596299112661746e-08f;       // log2 e - fp32(log2 e)
    const float u = p * kL;
    float e = __builtin_fmaf(p, kL, -u);
    e = __builtin_fmaf(p, kLlo, e);
    float ex = __builtin_amdgcn_exp2f(-u);
    ex = __builtin_fmaf(-ex, e * 0.693147182464599609375f, ex);
    const float large = 1.0f - ex;
    const float mag = t < 0.921875f ? small : large;
    return __builtin_copysignf(mag, z);
}

// ---- fast-class helpers for the remaining functors (16-bit I/O): hardware exp2/log2/rcp plus a short series where
// the closed form cancels.  Accuracy is checked for EVERY bf16 and fp16 input by tests/test_gpu_numerics.py.
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

// exp(x) - 1 for x <= 0 (the only side elu/celu/selu need; for x > 0 the caller's select discards the result,
// NaN propagates)
__device__ __forceinline__ float expm1_neg_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(x * kLog2e) - 1.0f;
    // |x| < 2^-6: x + x^2/2 + x^3/6 is exact to fp32 rounding; beyond it the subtraction loses < 2^-17 relative
    const float ser = __builtin_fmaf(__builtin_fmaf(x, 0.16666667f, 0.5f) * x, x, x);
    return x > -0.015625f ? ser : e;
}

// log(1 + e) for e >= 0
__device__ __forceinline__ float log1p_pos_fast(float e) {
    const float l = __builtin_amdgcn_logf(1.0f + e) * kLn2;          // v_log_f32 = log2
    const float ser = __builtin_fmaf(__builtin_fmaf(e, 0.33333334f, -0.5f) * e, e, e);
    return e < 0.0078125f ? ser : l;
}

// tanh(a) for a >= 0 as (1 - e)/(1 + e), e = exp(-2a) in (0, 1]: no overflow, NaN propagates, and the cancellation in
// 1 - e costs 2^-24/(2a) relative -- below 2^-14 from a = 2^-11 up, which is all a 16-bit result can see.
__device__ __forceinline__ float tanh_pos_fast(float a) {
    const float e = __builtin_amdgcn_exp2f(a * (-2.0f * kLog2e));
    return (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
}

__device__ __forceinline__ float tanh_fast(float x) {
    const float a = __builtin_fabsf(x);
    const float t = tanh_pos_fast(a);
    return __builtin_copysignf(a < 0x1p-11f ? a : t, x);       // tanh(a) = a (1 - a^2/3 + ...): a itself below 2^-11
}

// x - tanh(x): series x^3/3 - 2x^5/15 + 17x^7/315 - 62x^9/2835 + 1382 x^11/155925 below 1/2 (no cancellation),
// a - tanh(a) above (>= 0.038 there, so the 2^-24 absolute error of tanh is 2^-19 relative at worst)
__device__ __forceinline__ float tanhshrink_fast(float x) {
    const float a = __builtin_fabsf(x);
    const float s = a * a;
    float p = 0.0088632358f;
    p = __builtin_fmaf(p, s, -0.021869488f);
    p = __builtin_fmaf(p, s, 0.053968254f);
    p = __builtin_fmaf(p, s, -0.13333334f);
    p = __builtin_fmaf(p, s, 0.33333334f);
    const float small = p * s * a;
    const float big = a - tanh_pos_fast(a);
    return __builtin_copysignf(a < 0.5f ? small : big, x);
}

__device__ __forceinline__ float softplus_fast(float x, float beta, float threshold) {
    const float bx = x * beta;
    const float e = __builtin_amdgcn_exp2f(bx * kLog2e);
    return bx > threshold ? x : log1p_pos_fast(e) * __builtin_amdgcn_rcpf(beta);
}

__device__ __forceinline__ float logsigmoid_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(__builtin_fabsf(x) * -kLog2e);
    return __builtin_fminf(0.0f, x) - log1p_pos_fast(e);
}

// x * tanh(softplus(x)) = x * n / (n + 2),  n = e^x (e^x + 2): no cancellation anywhere; clamp keeps n finite
__device__ __forceinline__ float mish_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(__builtin_fminf(x, 30.0f) * kLog2e);
    const float n = e * (e + 2.0f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.0f));
}

template <int FN, bool FAST> struct Act {
    // p0/p1 are wave-uniform kernel arguments
    static __device__ __forceinline__ float eval(float x, float p0, float p1) {
        if constexpr (FN == FEWBIT_CELU) {
            if constexpr (FAST) return x > 0.0f ? x : p0 * expm1_neg_fast(x * __builtin_amdgcn_rcpf(p0));
            return x > 0.0f ? x : p0 * expm1f(x / p0);
        } else if constexpr (FN == FEWBIT_ELU) {
            if constexpr (FAST) return x > 0.0f ? x : p0 * expm1_neg_fast(x);
            return x > 0.0f ? x : p0 * expm1f(x);
        } else if constexpr (FN == FEWBIT_GELU) {
            if constexpr (FAST) return gelu_fast(x);
            // ATen: x * 0.5 * (1 + erf(x * M_SQRT1_2)), in this order, in fp32
            return (x * 0.5f) * (1.0f + erf_precise(x * 0.70710678118654752440f));
        } else if constexpr (FN == FEWBIT_HARDSWISH) {
            float t = fminf(fmaxf(x + 3.0f, 0.0f), 6.0f);
            return x * t / 6.0f;
        } else if constexpr (FN == FEWBIT_LOGSIGMOID) {
            if constexpr (FAST) return logsigmoid_fast(x);
            return fminf(0.0f, x) - log1pf(expf(-fabsf(x)));
        } else if constexpr (FN == FEWBIT_MISH) {
            if constexpr (FAST) return mish_fast(x);
            float sp = x > 20.0f ? x : log1pf(expf(x));
            return x * tanhf(sp);
        } else if constexpr (FN == FEWBIT_SELU) {
            const float alpha = 1.6732632423543772848170429916717f;
            const float scale = 1.0507009873554804934193349852946f;
            if constexpr (FAST) return x > 0.0f ? scale * x : (scale * alpha) * expm1_neg_fast(x);
            return x > 0.0f ? scale * x : (scale * alpha) * expm1f(x);
        } else if constexpr (FN == FEWBIT_SIGMOID) {
            if constexpr (FAST) return sigmoid_fast(x);
            return 1.0f / (1.0f + expf(-x));
        } else if constexpr (FN == FEWBIT_SILU) {
            if constexpr (FAST) return x * sigmoid_fast(x);
            return x / (1.0f + expf(-x));
        } else if constexpr (FN == FEWBIT_SOFTPLUS) {
            if constexpr (FAST) return softplus_fast(x, p0, p1);
            float bx = x * p0;
            return bx > p1 ? x : log1pf(expf(bx)) / p0;
        } else if constexpr (FN == FEWBIT_SOFTSIGN) {
            return x / (1.0f + fabsf(x));
        } else if constexpr (FN == FEWBIT_TANH) {
            if constexpr (FAST) return tanh_fast(x);
            return tanhf(x);
        } else if constexpr (FN == FEWBIT_TANHSHRINK) {
            if constexpr (FAST) return tanhshrink_fast(x);
            return x - tanhf(x);
        } else {
            return x;
        }
    }
    // the value the border table is searched with: x itself, except for the even-parity fold of a custom table
    // (FEWBIT_IDENTITY_FOLD, p0 = shift_x), where it is the fp32 distance from the shift
    static constexpr bool kFolded = (FN == FEWBIT_IDENTITY_FOLD);
    static __device__ __forceinline__ float key(float x, float p0) {
        if constexpr (kFolded) return fabsf(x - p0);
        return x;
    }
};

// 1-bit family: value and derivative-branch bit (fewbit/cuda/codec.cu:298-487 for the bit rules)
template <int FN> struct Step1 {
    static __device__ __forceinline__ float eval(float x, float p0, float p1, uint32_t &bit) {
        if constexpr (FN == FEWBIT_HARDSHRINK) {
            bool keep = (x < -p0) || (x > p0);
            bit = keep;
            return keep ? x : 0.0f;
        } else if constexpr (FN == FEWBIT_HARDSIGMOID) {
            bool lo = x <= -3.0f, hi = x >= 3.0f;
            bit = !(lo || hi);
            return lo ? 0.0f : (hi ? 1.0f : (x + 3.0f) / 6.0f);
        } else if constexpr (FN == FEWBIT_HARDTANH) {
            bool lo = x <= p0, hi = x >= p1;
            bit = !(lo || hi);
            return lo ? p0 : (hi ? p1 : x);
        } else if constexpr (FN == FEWBIT_LEAKY_RELU) {
            bool pos = x >= 0.0f;
            bit = !pos;
            return pos ? x : p0 * x;
        } else if constexpr (FN == FEWBIT_RELU) {
            bool off = x <= 0.0f;
            bit = !off;
            return off ? 0.0f : x;
        } else if constexpr (FN == FEWBIT_RELU6) {
            bool lo = x <= 0.0f, hi = x >= 6.0f;
            bit = !(lo || hi);
            return lo ? 0.0f : (hi ? 6.0f : x);
        } else if constexpr (FN == FEWBIT_SOFTSHRINK) {
            bool lo = x < -p0, hi = x > p0;
            bit = lo || hi;
            return lo ? x + p0 : (hi ? x - p0 : 0.0f);
        } else {  // FEWBIT_THRESHOLD
            bool off = x <= p0;
            bit = !off;
            return off ? p1 : x;
        }
    }
};

}  // namespace fewbit_hip
