// fewbit_dct.hip -- the sampled cosine transform of the randomized linear layers (the reference's 'dct' estimator)
// on gfx950:
//
//     out[j][:] = scale * DCT-II_ortho(M, along the rows)[idx[j]][:]        M: rows x features (bf16 / fp16 / fp32), rows = 2^m, 3 x 2^m or 5 x 2^m
//
// What it replaces in the reference (skolai/fewbit): `dct(input_view, dim=0, norm='ortho')[proj, ...]` in LinearGRPFunc.forward
// (fewbit/functional/linear.py:113-122) and the same on the gradient in .backward (:174-183); dct = fewbit/fft.py:10-43 (shuffle,
// torch.fft.fft along the transposed last dimension, phase multiply).  There the WHOLE transform is materialised in fp32 (for a
// 16-bit input after a cast) through a strided library FFT and the sampled rows are gathered afterwards: measured here 443 us /
// 1679 us for 16384 x 768 / 3072 bf16 (profiles/r06_sketch_bench.json), 110-120 x the bytes the result needs (read M once, write
// p rows).  This kernel pair moves M once, one fp32 intermediate once out and once back, and writes only the sampled rows.
//
// ---- the algorithm ---------------------------------------------------------------------------------------------------------
//   1. Makhoul's reordering (the reference's step 1, fewbit/fft.py:26): v[n] = x[2n] (n < N/2), v[N-1-n] = x[2n+1]; with
//      V = DFT_N(v):  DCT-II(x)[k] = Re(2 e^{-i pi k / 2N} V[k]).
//   2. Two real columns per complex transform: M is row-major, so features (2c, 2c+1) of a row ARE a complex number in memory;
//      Z = DFT_N(v_2c + i v_2c+1) gives V_2c[k] = (Z[k] + conj Z[N-k]) / 2 and V_2c+1[k] = (Z[k] - conj Z[N-k]) / 2i.
//   3. Four-step DFT, N = N1 x N2 (each 16 .. 512; 16384 = 128 x 128, 262144 = 512 x 512; 12288 = 128 x 96: a factor 3 goes to N2), n = N2 n1 + n2, k = k1 + N1 k2:
//          pass A   for every n2:  A[k1][n2] = W_N^{n2 k1} * sum_{n1} z[N2 n1 + n2] W_N1^{n1 k1}        (length-N1 DFTs over rows N2 apart)
//          pass B   for every k1:  Z[k1 + N1 k2] = sum_{n2} A[k1][n2] W_N2^{n2 k2}                       (length-N2 DFTs, contiguous)
//      Pass B never writes Z: the workgroup that owns the residues k1 and N1 - k1 holds Z[k] AND Z[N-k] for every k of those
//      two classes in LDS, scans idx for the samples that fall into them and writes just those rows of the result.
//
// ---- tiling ------------------------------------------------------------------------------------------------------------------
//   tile        L points x 32 complex fp32 entries = 32 KiB of LDS at L = 128 (+ 3-7 KiB of tables): FOUR 256-thread workgroups per CU
//               (L = 256, from 32768 rows on: 64 KiB, two per CU).
//               Lanes run along the 32 entries of a point: every LDS access of a half-wave is 256 contiguous bytes (all 64 banks once,
//               ds_read/write_b64: conflict-free), every twiddle is half-wave-uniform (an LDS broadcast).
//   FFT         in place, decimation in frequency, TWO stages at L = 128 (radix 16 then radix 8, each butterfly entirely in the
//               registers of one thread; 64 = 8 x 8, 32 = 8 x 4, 16 = 16; 96 = 3 x 8 x 4, 192 = 3 x 8 x 8, 48 = 3 x 16), one barrier per stage; the result stands in digit-reversed
//               positions (pos_to_freq / freq_to_pos), which costs nothing: both passes address their outputs through the map.
//   pass A      workgroup (b, t): the rows n = N2 n1 + b of column tile t (32 complex columns = 64 features).  Loads N1 row segments
//               (128 B of bf16: full cache lines, 16 B per lane, issued back to back), converts to fp32, transforms along n1,
//               multiplies by W_N^{n2 k1} (two table lookups, one complex multiply) and writes the intermediate per HALF tile,
//               [half tile][k1][n2][16 columns]: 128 contiguous bytes per (k1, half).
//   pass B      workgroup (u, t): residues k1 = u and N1 - u of half tile t (16 complex columns).  Its 32 entries per point are the two
//               residue rows side by side, so one butterfly serves both.  Reads two contiguous N2 x 128 B blocks, transforms along n2 and
//               writes the samples of its two residue classes: four lanes per sampled row (four complex columns each), 64 rows at a time,
//               e^{-i pi k / 2N} from two small tables (no transcendental per sample).
//   idx         an int64 array of the caller (fewbit_hip_sampled_dct, RowsInMemory) or a FUNCTION of a 64-bit seed (fewbit_hip_sampled_dct_seeded,
//               RowsOfSeed: Philox4x32-10) -- what the layer uses: no array, no launch that draws one, a seed to keep for backward, capturable
//               into a hipGraph.  Either way ONE workgroup of pass A sorts the samples by residue class into the workspace (sort_rows), so
//               that a pass-B workgroup reads exactly its own (each of them testing all of idx was a third of pass B's time).
//   traffic     M once + 2 x rows x features x 4 B of intermediate + the p sampled rows: 16384 x 768 bf16, p = 3276: 25 + 2 x 50 + 5 MB.
// Measured and not kept (round 6, profiles/r06_dct_variants.txt): 64 KiB tiles with 512-thread workgroups (same time); persistent
// workgroups that request the next tile before transforming the current one (the radix-16 butterfly leaves no registers for it: spills).
// Roofline class: HBM / Infinity Cache bandwidth (5 N log2 N flops per column: 0.9 GFLOP for 16384 x 768).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <type_traits>

#include "fewbit_hip.h"
#include "fewbit_philox.h"

#define FEWBIT_HIDDEN __attribute__((visibility("hidden")))

namespace fewbit_hip {

// shared with the core unit of fewbit_kernels.hip
FEWBIT_HIDDEN int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

namespace dct {

constexpr int C = 32;                       // complex columns of a tile = 64 features
constexpr int kFeatures = 2 * C;
// Both passes: a 32 KiB tile + tables per 256-thread workgroup, four workgroups per CU, each in its own phase (loading, transforming,
// storing): pass A one row of transforms x 32 complex columns; pass B the two rows of a residue pair x 16 complex columns
constexpr int kThreadsA = 256, kRowsA = 1;
constexpr int kThreadsB = 256, CB = C / 2;  // pass B: 16 complex columns x the two rows of a residue pair
constexpr int kServeLanes = 4;              // pass B: lanes that write one sampled row (CB / kServeLanes complex columns each)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// complex product with two multiplies and two fused multiply-adds (the build has -ffp-contract=off: fusion is spelled out where wanted)
__device__ __forceinline__ f32x2 cmul(f32x2 a, f32x2 b) {
    return f32x2{__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x)};
}
__device__ __forceinline__ f32x2 mul_mi(f32x2 a) { return f32x2{a.y, -a.x}; }                                   // a * (-i)
// e^{-2 pi i num / den}: den a power of two (num / den is exact in fp32), or 3 x / 5 x a power of two (the angle in double, then rounded;
// `den` is a template constant at every call site: the branch folds)
__device__ __forceinline__ f32x2 unit(int num, int den) {
    if ((den & (den - 1)) != 0) {
        double s, c;
        sincospi(-2.0 * static_cast<double>(num) / static_cast<double>(den), &s, &c);
        return f32x2{static_cast<float>(c), static_cast<float>(s)};
    }
    float s, c;
    sincospif(-2.0f * static_cast<float>(num) / static_cast<float>(den), &s, &c);
    return f32x2{c, s};
}

// ---- small transforms in registers, natural order in and out: x[q] <- sum_j x[j] e^{-2 pi i j q / R} ------------------------------
constexpr float kR2 = 0.70710678118654752f;                   // sqrt(1/2)
constexpr float kC8 = 0.92387953251128674f, kS8 = 0.38268343236508977f;      // cos, sin of pi / 8
__device__ __forceinline__ void dft2(f32x2 &a, f32x2 &b) {
    const f32x2 s = a + b, d = a - b;
    a = s;
    b = d;
}
__device__ __forceinline__ void dft4(f32x2 &a0, f32x2 &a1, f32x2 &a2, f32x2 &a3) {
    const f32x2 t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_mi(a1 - a3);
    a0 = t0 + t2;
    a1 = t1 + t3;
    a2 = t0 - t2;
    a3 = t1 - t3;
}
constexpr float kH3 = 0.86602540378443865f;                   // sqrt(3) / 2
template <int R> __device__ __forceinline__ void dft(f32x2 (&x)[R]) {
    if constexpr (R == 2) {
        dft2(x[0], x[1]);
    } else if constexpr (R == 3) {
        // W3 = -1/2 - i sqrt(3)/2:  y1, y2 = x0 - (x1 + x2) / 2  +-  (-i) (sqrt(3) / 2) (x1 - x2)
        const f32x2 t = x[1] + x[2], d = mul_mi(x[1] - x[2]) * kH3, m = x[0] - t * 0.5f;
        x[0] = x[0] + t;
        x[1] = m + d;
        x[2] = m - d;
    } else if constexpr (R == 4) {
        dft4(x[0], x[1], x[2], x[3]);
    } else if constexpr (R == 5) {
        // W5^j = cos(2 pi j / 5) - i sin(2 pi j / 5):  y1, y4 = m1 -+ i n1,  y2, y3 = m2 -+ i n2
        constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f, s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
        const f32x2 a1 = x[1] + x[4], a2 = x[2] + x[3], b1 = x[1] - x[4], b2 = x[2] - x[3];
        const f32x2 m1 = x[0] + a1 * c1 + a2 * c2, m2 = x[0] + a1 * c2 + a2 * c1;
        const f32x2 n1 = mul_mi(b1 * s1 + b2 * s2), n2 = mul_mi(b1 * s2 - b2 * s1);
        x[0] = x[0] + a1 + a2;
        x[1] = m1 + n1;
        x[4] = m1 - n1;
        x[2] = m2 + n2;
        x[3] = m2 - n2;
    } else if constexpr (R == 8) {
        // j = 2a + b, q = p + 4 q':  y[p + 4 q'] = sum_b W8^{bp} (-1)^{b q'} sum_a x[2a + b] W4^{ap}
        dft4(x[0], x[2], x[4], x[6]);
        dft4(x[1], x[3], x[5], x[7]);
        const f32x2 u1 = x[3], u2 = x[5], u3 = x[7];
        const f32x2 t0 = x[1];
        const f32x2 t1 = f32x2{(u1.x + u1.y) * kR2, (u1.y - u1.x) * kR2};             // * W8^1 = sqrt(1/2) (1 - i)
        const f32x2 t2 = mul_mi(u2);                                                   // * W8^2
        const f32x2 t3 = f32x2{(u3.y - u3.x) * kR2, -(u3.x + u3.y) * kR2};            // * W8^3 = -sqrt(1/2) (1 + i)
        const f32x2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
        x[0] = e0 + t0; x[4] = e0 - t0;
        x[1] = e1 + t1; x[5] = e1 - t1;
        x[2] = e2 + t2; x[6] = e2 - t2;
        x[3] = e3 + t3; x[7] = e3 - t3;
    } else {
        static_assert(R == 16, "radix 2, 3, 4, 5, 8 or 16");
        // j = 4a + b, q = p + 4 q':  y[p + 4 q'] = sum_b W16^{bp} W4^{b q'} sum_a x[4a + b] W4^{ap}
        dft4(x[0], x[4], x[8], x[12]);
        dft4(x[1], x[5], x[9], x[13]);
        dft4(x[2], x[6], x[10], x[14]);
        dft4(x[3], x[7], x[11], x[15]);
        // u[b][p] = x[4p + b]; twiddle W16^{bp}
        x[5] = cmul(x[5], f32x2{kC8, -kS8});                                           // b = 1, p = 1: W16^1
        x[9] = f32x2{(x[9].x + x[9].y) * kR2, (x[9].y - x[9].x) * kR2};               // b = 1, p = 2: W16^2 = W8^1
        x[13] = cmul(x[13], f32x2{kS8, -kC8});                                         // b = 1, p = 3: W16^3
        x[6] = f32x2{(x[6].x + x[6].y) * kR2, (x[6].y - x[6].x) * kR2};               // b = 2, p = 1: W16^2
        x[10] = mul_mi(x[10]);                                                         // b = 2, p = 2: W16^4 = -i
        x[14] = f32x2{(x[14].y - x[14].x) * kR2, -(x[14].x + x[14].y) * kR2};         // b = 2, p = 3: W16^6 = W8^3
        x[7] = cmul(x[7], f32x2{kS8, -kC8});                                           // b = 3, p = 1: W16^3
        x[11] = f32x2{(x[11].y - x[11].x) * kR2, -(x[11].x + x[11].y) * kR2};         // b = 3, p = 2: W16^6
        x[15] = cmul(x[15], f32x2{-kC8, kS8});                                         // b = 3, p = 3: W16^9
        // outer transforms over b for each p; result q' of group p is output p + 4 q' -- which is where dft4 leaves it when the
        // group is (x[4p], x[4p+1], x[4p+2], x[4p+3]) and the outputs are then transposed: y[p + 4q'] = group_p[q']
        dft4(x[0], x[1], x[2], x[3]);
        dft4(x[4], x[5], x[6], x[7]);
        dft4(x[8], x[9], x[10], x[11]);
        dft4(x[12], x[13], x[14], x[15]);
        f32x2 y[16];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) y[p + 4 * q] = x[4 * p + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = y[q];
    }
}

// radix of the first stage of a block of length `len`: 256 = 16 x 16, 128 = 16 x 8, 64 = 8 x 8, 32 = 8 x 4, 16 = 16; a factor 3
// (48 = 3 x 16, 96 = 3 x 8 x 4, 192 = 3 x 8 x 8) or 5 (80 = 5 x 16, 160 = 5 x 8 x 4) goes first
__host__ __device__ constexpr int first_radix(int len) {
    return len % 3 == 0 ? 3 : len % 5 == 0 ? 5 : len >= 128 ? 16 : len == 64 ? 8 : len == 32 ? 8 : len == 16 ? 16 : len == 8 ? 8 : len == 4 ? 4 : 2;
}

// position P (after the in-place DIF stages) -> frequency k.  Stage i with radix r_i on blocks of length L_i leaves digit q_i
// (k = q_1 + r_1 q_2 + r_1 r_2 q_3 + ...) in sub-block q_i: P = sum q_i L_i / r_i.
template <int LEN> __host__ __device__ __forceinline__ int pos_to_freq(int p) {
    if constexpr (LEN <= 1) {
        return 0;
    } else {
        constexpr int r = first_radix(LEN), s = LEN / r;
        return p / s + r * pos_to_freq<s>(p % s);
    }
}
template <int LEN> __host__ __device__ __forceinline__ int freq_to_pos(int k) {
    if constexpr (LEN <= 1) {
        return 0;
    } else {
        constexpr int r = first_radix(LEN), s = LEN / r;
        return (k % r) * s + freq_to_pos<s>(k / r);
    }
}

// One stage of the in-place transform of the tile [TR][L][C] along its middle axis: blocks of length LEN, radix R, twiddles
// tw[m] = W_L^m.  Thread (c = tid % 32, slot = tid / 32 of SLOTS) takes the butterflies slot, slot + SLOTS, ... of column c of all
// TR rows; a butterfly is R loads, the transform in registers, the twiddles W_LEN^{ss q} (none in the last stage) and R stores.
template <int L, int LEN, int R, int TR, int SLOTS> __device__ __forceinline__ void stage(f32x2 *tile, const f32x2 *tw, int c, int slot) {
    constexpr int S = LEN / R, kPerRow = L / R;
#pragma unroll
    for (int bid0 = 0; bid0 < TR * kPerRow; bid0 += SLOTS) {
        const int bid = bid0 + slot;
        if (TR * kPerRow % SLOTS != 0 && bid >= TR * kPerRow) break;
        const int row = bid / kPerRow, b = bid % kPerRow, block = b / S, ss = b % S;
        f32x2 *p = tile + (row * L + block * LEN + ss) * C + c;
        f32x2 x[R];
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = p[j * S * C];
        dft<R>(x);
        if constexpr (S > 1) {
#pragma unroll
            for (int q = 1; q < R; ++q) x[q] = cmul(x[q], tw[(L / LEN) * ss * q]);
        }
#pragma unroll
        for (int q = 0; q < R; ++q) p[q * S * C] = x[q];
    }
    __syncthreads();
}

template <int L, int TR, int SLOTS, int LEN = L> __device__ __forceinline__ void fft_tile(f32x2 *tile, const f32x2 *tw, int c, int slot) {
    if constexpr (LEN > 1) {
        constexpr int R = first_radix(LEN);
        stage<L, LEN, R, TR, SLOTS>(tile, tw, c, slot);
        fft_tile<L, TR, SLOTS, LEN / R>(tile, tw, c, slot);
    }
}

template <int DT> struct In {             // 16-byte piece of a row: 8 features of a 16-bit dtype, 4 of fp32
    static constexpr int kPieceFeatures = DT == FEWBIT_F32 ? 4 : 8;
    static constexpr int kPiecesPerSegment = kFeatures / kPieceFeatures;
};

__device__ __forceinline__ float half_to_float(uint32_t h, int dt) {
    if (dt == FEWBIT_BF16) return __builtin_bit_cast(float, h << 16);
    return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(h)));
}

// the piece `piece` of the 64-feature segment of row `row` that starts at feature f0, raw (16 bytes: 8 features of a 16-bit dtype or
// 4 of fp32).  FULL: the whole tile lies inside the matrix (a workgroup-uniform fact): one unguarded 16-byte load -- a load under a
// lane-divergent guard makes hipcc wait for it on the spot, which serialised the tile's loads; otherwise zeros beyond `features`,
// element by element.  The conversion to fp32 happens when the piece is written to LDS (unpack), after the latency has been used.
template <int DT, bool FULL> __device__ __forceinline__ u32x4 load_piece(const void *x, size_t row, size_t ld, size_t f0, int piece, size_t features) {
    constexpr int PF = In<DT>::kPieceFeatures;
    const size_t f = f0 + static_cast<size_t>(piece) * PF;
    if constexpr (DT == FEWBIT_F32) {
        const uint32_t *p = static_cast<const uint32_t *>(x) + row * ld + f;
        if (FULL || f + PF <= features) {
            typedef u32x4 __attribute__((aligned(4))) u32x4u;
            return *reinterpret_cast<const u32x4u *>(p);
        }
        u32x4 q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = f + e < features ? p[e] : 0u;
        return q;
    } else {
        const uint16_t *p = static_cast<const uint16_t *>(x) + row * ld + f;
        if (FULL || f + PF <= features) {
            typedef u32x4 __attribute__((aligned(2))) u32x4u;
            return *reinterpret_cast<const u32x4u *>(p);
        }
        u32x4 q = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e >> 1] |= (f + e < features ? static_cast<uint32_t>(p[e]) : 0u) << (16 * (e & 1));
        return q;
    }
}
template <int DT> __device__ __forceinline__ void unpack_piece(u32x4 q, float (&v)[In<DT>::kPieceFeatures]) {
    if constexpr (DT == FEWBIT_F32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t w = q[e];              // (a scalar first: __builtin_bit_cast applied to the element expression q[e] itself
            v[e] = __builtin_bit_cast(float, w);  //  reads the vector's first element for every e -- observed with hipcc 7.2)
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = half_to_float(q[e] & 0xffffu, DT);
            v[2 * e + 1] = half_to_float(q[e] >> 16, DT);
        }
    }
}

constexpr int kFine = 128;                  // W_D^e = fine[e % 128] * coarse[e / 128] for e < N (coarse: N / 128 entries, at most 2048)
constexpr int coarse_entries(int n) { return n / kFine > 0 ? n / kFine : 1; }
__device__ __forceinline__ f32x2 table_unit(const f32x2 *fine, const f32x2 *coarse, int e, bool has_coarse) {
    return has_coarse ? cmul(fine[e % kFine], coarse[e / kFine]) : fine[e % kFine];
}

// ---- the sampled rows ------------------------------------------------------------------------------------------------------------
// Where they come from.  RowsInMemory: the caller's int64 array.  RowsOfSeed: a FUNCTION of a 64-bit seed,
//     rows = 2^k:      idx[j] = 16-bit half j % 8 of the 128 bits of Philox4x32-10(counter = (j / 8, 0, 0, 3), key = seed)  mod  rows
//                      (half h = bits 16 (h % 2) .. 16 (h % 2) + 15 of word h / 2)
//     rows = 3 x 2^k:  idx[j] = (word j % 4 of Philox4x32-10(counter = (j / 4, 0, 0, 3), key = seed)  x  rows)  >>  32
// (uniform -- in the second case up to rows / 2^32 --, with replacement, like the reference's T.multinomial of equal weights): no array, no
// launch that draws one, nothing to keep for backward but the seed -- and, with the seed read from device memory, a recorded launch draws
// fresh rows on every replay (fewbit_sketch.hip, same scheme).
constexpr uint32_t kRowsDomain = 3u;        // counter word 3 (0 and 2: the dense sketches)
__host__ __device__ constexpr bool power_of_two(size_t n) { return (n & (n - 1)) == 0; }
// rows = 2^k <= 2^16: eight 16-bit halves per Philox call; any other row count (3 x 2^k, 5 x 2^k, 2^17, 2^18): four 32-bit words, word x rows >> 32
__host__ __device__ constexpr bool draws_halves(size_t n) { return power_of_two(n) && n <= 65536; }
__host__ __device__ constexpr int per_draw(bool halves) { return halves ? 8 : 4; }      // row numbers per Philox call
// row number h of one Philox call (HALVES: not yet reduced mod rows -- the caller masks)
template <bool HALVES> __host__ __device__ __forceinline__ int drawn_row(const uint32_t (&w)[4], int h, uint32_t rows) {
    if constexpr (HALVES) return static_cast<int>((w[h / 2] >> (16 * (h % 2))) & 0xffffu);
    else return static_cast<int>((static_cast<uint64_t>(w[h]) * rows) >> 32);
}
struct RowsInMemory {
    static constexpr bool kSeeded = false;
    const int64_t *idx;
};
struct RowsOfSeed {
    static constexpr bool kSeeded = true;
    sketch::Key value;
    const sketch::Key *device;              // != nullptr: the key is read from there when the kernel runs
};

// Pass B's workgroup (u, t) writes the samples k with k % N1 in {u, N1 - u}.  So that its 24 .. 96 siblings (one per half tile) need
// not each test all of idx -- a third of pass B's time when they did (profiles/r06_dct_variants.txt: "noenlist") -- ONE workgroup, pass
// A's (0, 0) before it turns to its own tile, sorts the samples by u into the workspace:
//     sorted[offsets[u] .. offsets[u + 1])  =  the samples (k, j) with min(k % N1, N1 - k % N1) = u,   u = 0 .. N1 / 2
// A counting sort in LDS: histogram, prefix, placement.  The order inside a class is whatever the atomics give; no result depends on it
// (every sample writes its own row of the output).
struct Sample { int k, j; };                // frequency, row of the output
constexpr size_t kOffsetsBytes = 2048;      // (N1 / 2 + 2 ints, N1 <= 512, rounded up)

template <int N1, int N, typename ROWS>
__device__ __forceinline__ void sort_rows(ROWS rows, size_t proj, int *__restrict__ offsets, Sample *__restrict__ sorted, int *lds, int tid) {
    constexpr int U = N1 / 2 + 1, kThreads = kThreadsA;
    constexpr bool kPow2 = power_of_two(N), kHalves = draws_halves(N);
    constexpr int kPerDraw = per_draw(kHalves);
    int *hist = lds, *cursor = lds + U + 1;
    sketch::Key key{0u, 0u};
    if constexpr (ROWS::kSeeded) {
        key = rows.value;
        if (rows.device != nullptr) key = *rows.device;               // (one scalar load)
    }
    auto bucket = [](int k) -> int {
        const int k1 = k % N1;
        return k1 <= N1 / 2 ? k1 : N1 - k1;
    };
    // every (k, j) this thread looks after: of an array j = tid, tid + 256, ...; of a seed the numbers of the Philox calls tid, tid + 256, ...
    auto for_each = [&](auto &&f) __attribute__((always_inline)) {
        if constexpr (ROWS::kSeeded) {
            for (size_t q = tid; kPerDraw * q < proj; q += kThreads) {
                uint32_t w[4];
                sketch::philox4x32(static_cast<uint32_t>(q), static_cast<uint32_t>(q >> 32), 0u, kRowsDomain, key, w);
#pragma unroll
                for (int h = 0; h < kPerDraw; ++h) {
                    const size_t j = kPerDraw * q + h;
                    const int drawn = drawn_row<kHalves>(w, h, N);
                    if (j < proj) f(kHalves ? drawn & (N - 1) : drawn, j);
                }
            }
        } else {
            // (the low dword of an int64 in [0, N) is the number; whatever else the array holds is reduced to [0, N))
            // eight unconditional requests (a clamped index) before the first is looked at: eight latencies overlap instead of following one another
            const int *lo = reinterpret_cast<const int *>(rows.idx);
            for (size_t j0 = tid; j0 < proj; j0 += 8 * kThreads) {
                int word[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const size_t j = j0 + static_cast<size_t>(a) * kThreads;
                    word[a] = lo[2 * (j < proj ? j : proj - 1)];
                }
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const size_t j = j0 + static_cast<size_t>(a) * kThreads;
                    if (j < proj) f(kPow2 ? word[a] & (N - 1) : static_cast<int>(static_cast<unsigned>(word[a]) % static_cast<unsigned>(N)), j);
                }
            }
        }
    };
    for (int b = tid; b <= U; b += kThreads) hist[b] = 0;
    __syncthreads();
    for_each([&](int k, size_t) { atomicAdd(&hist[bucket(k)], 1); });
    __syncthreads();
    for (int mine = tid; mine <= U; mine += kThreads) {                // offsets[b] = the classes before b (every lane reads the same word: a broadcast)
        int run = 0;
#pragma unroll 8
        for (int b = 0; b < U; ++b) {
            const int n = hist[b];
            run += b < mine ? n : 0;
        }
        cursor[mine] = run;
        offsets[mine] = run;
    }
    __syncthreads();
    for_each([&](int k, size_t j) {
        const int pos = atomicAdd(&cursor[bucket(k)], 1);
        sorted[pos] = Sample{k, static_cast<int>(j)};
    });
    __syncthreads();                                                   // (the tile takes this LDS over)
}

// ---- pass A -----------------------------------------------------------------------------------------------------------------
// grid (N2, column tiles): workgroup (b, t) transforms the rows n = N2 n1 + b of column tile t.  inter: [half tile][k1][n2][16] complex fp32.
template <int DT, int N1, int N2, typename ROWS>
__global__ __launch_bounds__(kThreadsA, (N1 > 128 ? 2 : 4)) void dct_pass_a_kernel(const void *__restrict__ x, size_t features, size_t ld, f32x2 *__restrict__ inter, ROWS rows,
                                                                  size_t proj, int *__restrict__ offsets, Sample *__restrict__ sorted) {
    constexpr int N = N1 * N2, kCoarse = coarse_entries(N), kThreads = kThreadsA, kSlots = kThreads / C;
    static_assert(kRowsA == 1, "one n2 per workgroup");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    f32x2 *tile = reinterpret_cast<f32x2 *>(lds_raw);                 // [N1][C]
    f32x2 *tw = tile + N1 * C;                                        // W_N1^m, m < N1
    f32x2 *fine = tw + N1, *coarse = fine + kFine;                    // W_N^m (m < 128), W_N^{128 m}
    const int tid = threadIdx.x, c = tid % C, slot = tid / C;
    const int b = blockIdx.x;
    const size_t t = blockIdx.y, f0 = t * kFeatures;
    // one workgroup of the launch sorts the sampled rows for pass B first (~2 us of its own time; the launch takes 16)
    if (blockIdx.x == 0 && blockIdx.y == 0) sort_rows<N1, N, ROWS>(rows, proj, offsets, sorted, reinterpret_cast<int *>(lds_raw), tid);

    // ---- loads first (all in flight), tables while they travel, then registers -> LDS
    constexpr int PF = In<DT>::kPieceFeatures, PPS = In<DT>::kPiecesPerSegment, kTotal = N1 * PPS, kPieces = (kTotal + kThreads - 1) / kThreads;
    // (one body per case, FULL tile or edge tile, each with its own registers from the loads to the LDS writes: were the two
    // cases to meet in one set of registers in between, the copies at the join would wait for the loads right behind their issue)
    auto fill_tile = [&](auto full) __attribute__((always_inline)) {
        u32x4 raw[kPieces];
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int pid = tid + kThreads * i, n1 = pid / PPS, piece = pid % PPS;
            if (kTotal % kThreads != 0 && pid >= kTotal) break;
            const int n = N2 * n1 + b;                                  // index into the reordered sequence v
            const size_t row = n < N / 2 ? 2 * static_cast<size_t>(n) : 2 * static_cast<size_t>(N - 1 - n) + 1;
            raw[i] = load_piece<DT, decltype(full)::value>(x, row, ld, f0, piece, features);
        }
        for (int m = tid; m < N1; m += kThreads) tw[m] = unit(m, N1);
        for (int m = tid; m < kFine; m += kThreads) fine[m] = unit(m, N);
        for (int m = tid; m < kCoarse; m += kThreads) coarse[m] = unit(m * kFine, N);
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int pid = tid + kThreads * i, n1 = pid / PPS, piece = pid % PPS;
            if (kTotal % kThreads != 0 && pid >= kTotal) break;
            float v[PF];
            unpack_piece<DT>(raw[i], v);
            f32x4 *dst = reinterpret_cast<f32x4 *>(tile + n1 * C + piece * (PF / 2));
#pragma unroll
            for (int e = 0; e < PF / 4; ++e) dst[e] = f32x4{v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]};
        }
    };
    if (f0 + kFeatures <= features) fill_tile(std::true_type{});       // (workgroup-uniform)
    else fill_tile(std::false_type{});
    __syncthreads();

    fft_tile<N1, kRowsA, kSlots>(tile, tw, c, slot);

    // ---- twiddle + store: unit = two complex columns (16 B) of one position P; 16 consecutive lanes = the 256 contiguous bytes of one k1
    constexpr int kUnitsTotal = N1 * (C / 2), kUnits = (kUnitsTotal + kThreads - 1) / kThreads;
#pragma unroll
    for (int i = 0; i < kUnits; ++i) {
        const int uid = tid + kThreads * i, c2 = uid % (C / 2), p = uid / (C / 2);
        if (kUnitsTotal % kThreads != 0 && uid >= kUnitsTotal) break;
        const int k1 = pos_to_freq<N1>(p), e = b * k1;
        const f32x2 w = table_unit(fine, coarse, e, kCoarse > 1);
        const f32x4 z = *reinterpret_cast<const f32x4 *>(tile + p * C + 2 * c2);
        const f32x2 a = cmul(f32x2{z[0], z[1]}, w), bb = cmul(f32x2{z[2], z[3]}, w);
        // (the intermediate is kept per HALF tile of 16 complex columns -- what one pass-B workgroup reads: 128 contiguous bytes here)
        f32x4 *dst = reinterpret_cast<f32x4 *>(inter + (((2 * t + c2 / (CB / 2)) * N1 + k1) * N2 + b) * CB + 2 * (c2 % (CB / 2)));
        *dst = f32x4{a.x, a.y, bb.x, bb.y};
    }
}

// ---- pass B -----------------------------------------------------------------------------------------------------------------
// grid (N1 / 2 + 1, half tiles of 16 complex columns): residues k1 = u and (N1 - u) % N1.  The LDS tile is [N2][2][16]: the two
// residue rows sit side by side, so that a half-wave still touches 256 contiguous bytes per position and one butterfly serves
// both rows -- to the transform it is one row of 32 columns.
template <int DT, int N1, int N2>
__global__ __launch_bounds__(kThreadsB, (N2 > 128 ? 2 : 4)) void dct_pass_b_kernel(const f32x2 *__restrict__ inter, const int *__restrict__ offsets, const Sample *__restrict__ sorted,
                                                                  size_t proj, size_t features, float scale, void *__restrict__ out) {
    constexpr int N = N1 * N2, kCoarse = coarse_entries(N), kThreads = kThreadsB, kSlots = kThreads / C;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    f32x2 *tile = reinterpret_cast<f32x2 *>(lds_raw);                 // [N2][2][CB]
    f32x2 *tw = tile + N2 * 2 * CB;                                   // W_N2^m
    f32x2 *fine = tw + N2, *coarse = fine + kFine;                    // W_4N^m (m < 128), W_4N^{128 m}: e^{-i pi k / 2N} = W_4N^k
    const int tid = threadIdx.x;
    const int u = blockIdx.x, k1a = u, k1b = (N1 - u) % N1;
    const size_t t = blockIdx.y, f0 = t * (2 * CB);
    // four lanes write one sampled row, four complex columns each: 64 rows at a time -- all of a typical workgroup's in one step (with 16
    // lanes per row the four dependent steps of sample -> tile -> table -> store cost 1.5 us of a 18.8 us launch, profiles/r06_dct_serve_lanes.txt)
    constexpr int kLanes = kServeLanes, E = CB / kLanes, kSGroups = kThreads / kLanes;       // lanes per sample, complex columns per lane
    const int lane = tid % kLanes, sgroup = tid / kLanes;

    // this workgroup's samples (pass A's workgroup (0, 0) sorted them): the first two of every group of lanes are requested FIRST (vmcnt
    // counts in order: looking at them after the transform does not wait for anything), unconditionally (a clamped index: hipcc gives a
    // load under a lane-divergent guard its own wait)
    const int begin = offsets[u], count = offsets[u + 1] - begin;      // (scalar loads)
    constexpr int kPre = 2;
    Sample pre[kPre];
#pragma unroll
    for (int i = 0; i < kPre; ++i) {
        const size_t e = static_cast<size_t>(begin) + sgroup + i * kSGroups;
        pre[i] = sorted[e < proj ? e : proj - 1];
    }
    constexpr int kPerRow = N2 * (CB / 2), kTotal = 2 * kPerRow, kPieces = (kTotal + kThreads - 1) / kThreads;     // 16-byte pieces (two complex)
    f32x4 v[kPieces];
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int pid = tid + kThreads * i, r = pid / kPerRow, rest = pid % kPerRow;
        if (kTotal % kThreads != 0 && pid >= kTotal) break;
        const f32x2 *src = inter + (t * N1 + (r == 0 ? k1a : k1b)) * static_cast<size_t>(N2) * CB;
        v[i] = reinterpret_cast<const f32x4 *>(src)[rest];
    }
    for (int m = tid; m < N2; m += kThreads) tw[m] = unit(m, N2);
    for (int m = tid; m < kFine; m += kThreads) fine[m] = unit(m, 4 * N);
    for (int m = tid; m < kCoarse; m += kThreads) coarse[m] = unit(m * kFine, 4 * N);
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int pid = tid + kThreads * i, r = pid / kPerRow, rest = pid % kPerRow, n2 = rest / (CB / 2), c2 = rest % (CB / 2);
        if (kTotal % kThreads != 0 && pid >= kTotal) break;
        *reinterpret_cast<f32x4 *>(tile + (n2 * 2 + r) * CB + 2 * c2) = v[i];
    }
    __syncthreads();
    fft_tile<N2, 1, kSlots>(tile, tw, tid % C, tid / C);               // (ends with a barrier)

    // ---- the sampled rows of this workgroup's two residue classes: one per group of kServeLanes lanes at a time, lanes along the columns
    const float base = scale * __builtin_sqrtf(0.5f / static_cast<float>(N));          // ortho: sqrt(1 / 2N) (k > 0), sqrt(1 / 4N) (k = 0)
    auto write_row = [&](int km, size_t j) __attribute__((always_inline)) {
        const int k1 = km % N1, k2 = km / N1;
        const int r = k1 == k1a ? 0 : 1;
        const int k2m = k1 == 0 ? (N2 - k2) % N2 : N2 - 1 - k2;         // N - k = (N1 - k1) + N1 k2m
        const f32x2 *pk = tile + (freq_to_pos<N2>(k2) * 2 + r) * CB + E * lane;
        const f32x2 *pm = tile + (freq_to_pos<N2>(k2m) * 2 + (1 - r)) * CB + E * lane;
        f32x2 zk[E], zm[E];
        if constexpr (E >= 2) {
#pragma unroll
            for (int e = 0; e < E; e += 2) {
                const f32x4 a4 = *reinterpret_cast<const f32x4 *>(pk + e), b4 = *reinterpret_cast<const f32x4 *>(pm + e);
                zk[e] = f32x2{a4[0], a4[1]}; zk[e + 1] = f32x2{a4[2], a4[3]};
                zm[e] = f32x2{b4[0], b4[1]}; zm[e + 1] = f32x2{b4[2], b4[3]};
            }
        } else {
            zk[0] = pk[0]; zm[0] = pm[0];
        }
        const f32x2 w = table_unit(fine, coarse, km, kCoarse > 1);      // e^{-i pi k / 2N}
        const float f = (km == 0 ? kR2 : 1.0f) * 2.0f * base;
        float y[2 * E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            f32x2 m = zm[e];
            m.y = -m.y;                                                 // conj Z[N - k]
            const f32x2 va = (zk[e] + m) * 0.5f, d = (zk[e] - m) * 0.5f, vb = mul_mi(d);
            y[2 * e] = (w.x * va.x - w.y * va.y) * f;                   // Re(w V)
            y[2 * e + 1] = (w.x * vb.x - w.y * vb.y) * f;
        }
        const size_t fa = f0 + 2 * E * lane;
        if constexpr (DT == FEWBIT_F32) {
            float *o = static_cast<float *>(out) + j * features + fa;
            if (fa + 2 * E <= features) {
                if constexpr (E == 1) {
                    *reinterpret_cast<f32x2 *>(o) = f32x2{y[0], y[1]};
                } else {
                    typedef f32x4 __attribute__((aligned(4))) f32x4u;
#pragma unroll
                    for (int e = 0; e < E; e += 2) *reinterpret_cast<f32x4u *>(o + 2 * e) = f32x4{y[2 * e], y[2 * e + 1], y[2 * e + 2], y[2 * e + 3]};
                }
            } else {
#pragma unroll
                for (int e = 0; e < 2 * E; ++e) if (fa + e < features) o[e] = y[e];
            }
        } else {
            uint16_t *o = static_cast<uint16_t *>(out) + j * features + fa;
            uint16_t h[2 * E];
#pragma unroll
            for (int e = 0; e < 2 * E; ++e) {
                if constexpr (DT == FEWBIT_BF16) h[e] = __builtin_bit_cast(uint16_t, static_cast<__bf16>(y[e]));
                else h[e] = __builtin_bit_cast(uint16_t, static_cast<_Float16>(y[e]));
            }
            if (fa + 2 * E <= features) {
                typedef uint32_t __attribute__((aligned(2))) u32u;
                typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
                typedef u32x2 __attribute__((aligned(2))) u32x2u;
                typedef u32x4 __attribute__((aligned(2))) u32x4u;
                auto pair = [&](int e) -> uint32_t { return static_cast<uint32_t>(h[2 * e]) | (static_cast<uint32_t>(h[2 * e + 1]) << 16); };
                if constexpr (E == 1) *reinterpret_cast<u32u *>(o) = pair(0);
                else if constexpr (E == 2) *reinterpret_cast<u32x2u *>(o) = u32x2{pair(0), pair(1)};
                else *reinterpret_cast<u32x4u *>(o) = u32x4{pair(0), pair(1), pair(2), pair(3)};
            } else {
#pragma unroll
                for (int e = 0; e < 2 * E; ++e) if (fa + e < features) o[e] = h[e];
            }
        }
    };
#pragma unroll
    for (int i = 0; i < kPre; ++i)
        if (sgroup + i * kSGroups < count) write_row(pre[i].k, static_cast<size_t>(pre[i].j));
    // (more than 128 samples in these two classes: p beyond ~8000, or a skewed idx)
    for (int e = sgroup + kPre * kSGroups; e < count; e += kSGroups) {
        const Sample smp = sorted[static_cast<size_t>(begin) + e];
        write_row(smp.k, static_cast<size_t>(smp.j));
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------------
struct Split { int n1, n2; };
// rows = N1 x N2.  2^8 .. 2^18: 16 <= N2 <= N1 <= 512, both powers of two (2^17 = 512 x 256 and 2^18 = 512 x 512: 128 KiB tiles, one
// workgroup per CU).  3 x 2^8 .. 3 x 2^14 (768 .. 49152) and 5 x 2^8 .. 5 x 2^13 (1280 .. 40960): the odd factor goes to the second pass,
// N2 = 48 / 96 / 192 or 80 / 160 (N1 stays a power of two: residues and digit maps of pass A, the k % N1 of pass B)
inline bool split_rows(size_t rows, Split &s) {
    if (rows < 256 || rows > 262144) return false;
    const bool three = rows % 3 == 0, five = !three && rows % 5 == 0;
    const size_t two = three ? rows / 3 : five ? rows / 5 : rows;
    if (!power_of_two(two) || (three && (two < 256 || two > 16384)) || (five && (two < 256 || two > 8192))) return false;
    int bits = 0;
    while ((static_cast<size_t>(1) << bits) < two) ++bits;
    if (!three && !five) {
        s.n1 = 1 << ((bits + 1) / 2);
        s.n2 = 1 << (bits / 2);
        return true;
    }
    // 3 x 2^bits, bits = 8 .. 14:  16 x 48, 32 x 48, 32 x 96, 64 x 96, 128 x 96, 128 x 192, 256 x 192
    // 5 x 2^bits, bits = 8 .. 13:  16 x 80, 32 x 80, 64 x 80, 64 x 160, 128 x 160, 256 x 160
    static const int n1_of_3[7] = {16, 32, 32, 64, 128, 128, 256}, n1_of_5[6] = {16, 32, 64, 64, 128, 256};
    s.n1 = three ? n1_of_3[bits - 8] : n1_of_5[bits - 8];
    s.n2 = static_cast<int>(rows / static_cast<size_t>(s.n1));
    return true;
}
inline size_t tiles_of(size_t features) { return (features + kFeatures - 1) / kFeatures; }
inline size_t inter_bytes(size_t rows, size_t features) { return tiles_of(features) * rows * C * sizeof(f32x2); }
// workspace: [the intermediate | offsets of the sorted samples (2 KiB) | the sorted samples, 8 bytes each]
inline size_t workspace_bytes_of(size_t rows, size_t features, size_t proj) { return inter_bytes(rows, features) + kOffsetsBytes + ((proj * sizeof(Sample) + 15) & ~static_cast<size_t>(15)); }

template <int L> constexpr size_t lds_bytes_a(int n) { return (kRowsA * L * C + L + kFine + coarse_entries(n)) * sizeof(f32x2); }
template <int L> constexpr size_t lds_bytes_b(int n) { return (2 * L * CB + L + kFine + coarse_entries(n)) * sizeof(f32x2); }

template <typename K> int opt_in(K kern, size_t lds, std::atomic<unsigned long long> &done) {
    if (lds <= 65536) return FEWBIT_OK;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_relaxed) & bit)) {                 // (once per kernel and device)
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
            (void)hipGetLastError();
            return fail(FEWBIT_ERR_LAUNCH, "sampled_dct: cannot reserve %zu bytes of LDS", lds);
        }
        done.fetch_or(bit, std::memory_order_relaxed);
    }
    return FEWBIT_OK;
}

template <int DT, int N1, int N2, typename ROWS>
int launch(const void *m, size_t features, size_t ld, ROWS idx, size_t proj, float scale, void *out, f32x2 *inter, int *offsets, Sample *sorted, hipStream_t s) {
    static std::atomic<unsigned long long> done_a{0}, done_b{0};
    constexpr size_t la = lds_bytes_a<N1>(N1 * N2), lb = lds_bytes_b<N2>(N1 * N2);
    static_assert(la >= 2 * (N1 / 2 + 2) * sizeof(int) && (N1 / 2 + 2) * sizeof(int) <= kOffsetsBytes, "the sort's counters fit pass A's LDS and the offsets their slot");
    if (const int rc = opt_in(dct_pass_a_kernel<DT, N1, N2, ROWS>, la, done_a)) return rc;
    if (const int rc = opt_in(dct_pass_b_kernel<DT, N1, N2>, lb, done_b)) return rc;
    const unsigned tiles = static_cast<unsigned>(tiles_of(features));
    hipLaunchKernelGGL((dct_pass_a_kernel<DT, N1, N2, ROWS>), dim3(N2, tiles), dim3(kThreadsA), la, s, m, features, ld, inter, idx, proj, offsets, sorted);
    const unsigned half_tiles = static_cast<unsigned>((features + 2 * CB - 1) / (2 * CB));
    hipLaunchKernelGGL((dct_pass_b_kernel<DT, N1, N2>), dim3(N1 / 2 + 1, half_tiles), dim3(kThreadsB), lb, s, inter, offsets, sorted, proj, features, scale, out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FEWBIT_ERR_LAUNCH, "sampled_dct: %s", hipGetErrorString(e));
    return FEWBIT_OK;
}

template <int DT, typename ROWS>
int launch_rows(Split sp, const void *m, size_t features, size_t ld, ROWS idx, size_t proj, float scale, void *out, f32x2 *inter, int *offsets, Sample *sorted, hipStream_t s) {
#define FB_DCT_CASE(A, B) \
    if (sp.n1 == A && sp.n2 == B) return launch<DT, A, B, ROWS>(m, features, ld, idx, proj, scale, out, inter, offsets, sorted, s);
    FB_DCT_CASE(16, 16) FB_DCT_CASE(32, 16) FB_DCT_CASE(32, 32) FB_DCT_CASE(64, 32) FB_DCT_CASE(64, 64) FB_DCT_CASE(128, 64) FB_DCT_CASE(128, 128)
    FB_DCT_CASE(256, 128) FB_DCT_CASE(256, 256) FB_DCT_CASE(512, 256) FB_DCT_CASE(512, 512)
    FB_DCT_CASE(16, 48) FB_DCT_CASE(32, 48) FB_DCT_CASE(32, 96) FB_DCT_CASE(64, 96) FB_DCT_CASE(128, 96) FB_DCT_CASE(128, 192) FB_DCT_CASE(256, 192)
    FB_DCT_CASE(16, 80) FB_DCT_CASE(32, 80) FB_DCT_CASE(64, 80) FB_DCT_CASE(64, 160) FB_DCT_CASE(128, 160) FB_DCT_CASE(256, 160)
#undef FB_DCT_CASE
    return fail(FEWBIT_ERR_UNSUPPORTED, "sampled_dct: no kernel for %d x %d rows", sp.n1, sp.n2);
}

// The file is compiled as three translation units in parallel, one per dtype (-DFEWBIT_DCT_TU=0 / 1 / 2 = FEWBIT_F32 / F16 / BF16: 54
// kernels each; the C entry points with unit 0), and linked into the one library -- 50 s instead of 2.5 min; without the define everything
// is one unit (make variant).
#ifndef FEWBIT_DCT_TU
#define FEWBIT_DCT_TU -1
#endif
#define FB_DCT_LAUNCH_ROWS(KEYWORD, DT, ROWS) \
    KEYWORD template int launch_rows<DT, ROWS>(Split, const void *, size_t, size_t, ROWS, size_t, float, void *, f32x2 *, int *, Sample *, hipStream_t);
#if FEWBIT_DCT_TU >= 0
FB_DCT_LAUNCH_ROWS(, FEWBIT_DCT_TU, RowsInMemory) FB_DCT_LAUNCH_ROWS(, FEWBIT_DCT_TU, RowsOfSeed)
#endif
#if FEWBIT_DCT_TU == 0
FB_DCT_LAUNCH_ROWS(extern, FEWBIT_F16, RowsInMemory) FB_DCT_LAUNCH_ROWS(extern, FEWBIT_F16, RowsOfSeed)
FB_DCT_LAUNCH_ROWS(extern, FEWBIT_BF16, RowsInMemory) FB_DCT_LAUNCH_ROWS(extern, FEWBIT_BF16, RowsOfSeed)
#endif
#undef FB_DCT_LAUNCH_ROWS

#if FEWBIT_DCT_TU <= 0
template <typename ROWS>
int run(int dtype, const void *m, size_t rows, size_t features, size_t ld, ROWS idx, size_t proj, double scale, void *out, void *workspace,
                       size_t workspace_bytes, void *stream) {
    Split sp;
    if (!split_rows(rows, sp)) return fail(FEWBIT_ERR_UNSUPPORTED, "sampled_dct: rows = %zu is none of 2^k (256 .. 262144), 3 x 2^k (768 .. 49152), 5 x 2^k (1280 .. 40960)", rows);
    if (m == nullptr || out == nullptr) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_dct: null pointer");
    if (ld < features) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_dct: leading dimension %zu < features %zu", ld, features);
    const size_t need = workspace_bytes_of(rows, features, proj);
    if (workspace == nullptr || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
        return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_dct: a 16-byte aligned workspace of %zu bytes is needed (fewbit_hip_sampled_dct_workspace), got %zu", need, workspace_bytes);
    if (tiles_of(features) > 32767) return fail(FEWBIT_ERR_UNSUPPORTED, "sampled_dct: more than 32767 column tiles");
    if (proj > 0x7fffffffull) return fail(FEWBIT_ERR_UNSUPPORTED, "sampled_dct: more than 2^31 - 1 samples");
    hipStream_t s = static_cast<hipStream_t>(stream);
    f32x2 *inter = static_cast<f32x2 *>(workspace);
    int *offsets = reinterpret_cast<int *>(static_cast<uint8_t *>(workspace) + inter_bytes(rows, features));
    Sample *sorted = reinterpret_cast<Sample *>(reinterpret_cast<uint8_t *>(offsets) + kOffsetsBytes);
    const float fs = static_cast<float>(scale);
    switch (dtype) {
    case FEWBIT_F32: return launch_rows<FEWBIT_F32, ROWS>(sp, m, features, ld, idx, proj, fs, out, inter, offsets, sorted, s);
    case FEWBIT_F16: return launch_rows<FEWBIT_F16, ROWS>(sp, m, features, ld, idx, proj, fs, out, inter, offsets, sorted, s);
    default: return launch_rows<FEWBIT_BF16, ROWS>(sp, m, features, ld, idx, proj, fs, out, inter, offsets, sorted, s);
    }
}

#endif  // FEWBIT_DCT_TU <= 0

}  // namespace dct
}  // namespace fewbit_hip

#if FEWBIT_DCT_TU <= 0
using namespace fewbit_hip;
using namespace fewbit_hip::dct;

extern "C" {

size_t fewbit_hip_sampled_dct_workspace(int dtype, size_t rows, size_t features, size_t proj) {
    Split sp;
    (void)dtype;
    if (features == 0 || proj == 0 || !split_rows(rows, sp)) return 0;
    return workspace_bytes_of(rows, features, proj);
}

int fewbit_hip_sampled_dct(int dtype, const void *m, size_t rows, size_t features, size_t ld, const int64_t *idx, size_t proj, double scale, void *out,
                           void *workspace, size_t workspace_bytes, void *stream) {
    if (dtype != FEWBIT_F32 && dtype != FEWBIT_F16 && dtype != FEWBIT_BF16) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_dct: unknown dtype %d", dtype);
    if (proj == 0 || features == 0) return FEWBIT_OK;
    if (idx == nullptr) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_dct: null pointer");
    return run(dtype, m, rows, features, ld, RowsInMemory{idx}, proj, scale, out, workspace, workspace_bytes, stream);
}

int fewbit_hip_sampled_dct_seeded(int dtype, const void *m, size_t rows, size_t features, size_t ld, uint64_t seed, const uint64_t *seed_device, size_t proj,
                                  double scale, void *out, void *workspace, size_t workspace_bytes, void *stream) {
    if (dtype != FEWBIT_F32 && dtype != FEWBIT_F16 && dtype != FEWBIT_BF16) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_dct: unknown dtype %d", dtype);
    if (proj == 0 || features == 0) return FEWBIT_OK;
    if ((reinterpret_cast<uintptr_t>(seed_device) & 7) != 0) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_dct: the seed word in device memory must be 8-byte aligned");
    const RowsOfSeed of{sketch::Key{static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)}, reinterpret_cast<const sketch::Key *>(seed_device)};
    return run(dtype, m, rows, features, ld, of, proj, scale, out, workspace, workspace_bytes, stream);
}

int fewbit_hip_sampled_rows(uint64_t seed, size_t rows, size_t proj, int64_t *idx) {
    Split sp;
    if (!split_rows(rows, sp)) return fail(FEWBIT_ERR_UNSUPPORTED, "sampled_rows: rows = %zu is none of 2^k (256 .. 262144), 3 x 2^k (768 .. 49152), 5 x 2^k (1280 .. 40960)", rows);
    if (proj > 0 && idx == nullptr) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_rows: null pointer");
    const sketch::Key key{static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)};
    const bool pow2 = draws_halves(rows);
    const size_t per = per_draw(pow2);
    for (size_t q = 0; per * q < proj; ++q) {
        uint32_t w[4];
        sketch::philox4x32(static_cast<uint32_t>(q), static_cast<uint32_t>(q >> 32), 0u, kRowsDomain, key, w);
        for (size_t h = 0; h < per && per * q + h < proj; ++h)
            idx[per * q + h] = pow2 ? static_cast<int64_t>(drawn_row<true>(w, static_cast<int>(h), 0u) & (rows - 1))
                                    : static_cast<int64_t>(drawn_row<false>(w, static_cast<int>(h), static_cast<uint32_t>(rows)));
    }
    return FEWBIT_OK;
}

}  // extern "C"
#endif  // FEWBIT_DCT_TU <= 0
