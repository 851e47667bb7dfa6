// fewbit_dct.hip -- the sampled cosine transform of the randomized linear layers (the reference's 'dct' estimator)
// on gfx950:
//
//     out[j][:] = scale * DCT-II_ortho(M, along the rows)[idx[j]][:]        M: rows x features (bf16 / fp16 / fp32), rows = 2^m
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
//   3. Four-step DFT, N = N1 x N2 (each 16 .. 128; 16384 = 128 x 128), n = N2 n1 + n2, k = k1 + N1 k2:
//          pass A   for every n2:  A[k1][n2] = W_N^{n2 k1} * sum_{n1} z[N2 n1 + n2] W_N1^{n1 k1}        (length-N1 DFTs over rows N2 apart)
//          pass B   for every k1:  Z[k1 + N1 k2] = sum_{n2} A[k1][n2] W_N2^{n2 k2}                       (length-N2 DFTs, contiguous)
//      Pass B never writes Z: the workgroup that owns the residues k1 and N1 - k1 holds Z[k] AND Z[N-k] for every k of those
//      two classes in LDS, scans idx for the samples that fall into them and writes just those rows of the result.
//
// ---- tiling ------------------------------------------------------------------------------------------------------------------
//   tile        2 rows-of-transforms x L points x 32 complex columns (64 features) of fp32 complex = 64 KiB of LDS at L = 128, two
//               workgroups per CU.  Lanes run along the columns: every LDS access of a half-wave is 256 contiguous bytes (all 64
//               banks once, ds_read/write_b64: conflict-free), every twiddle is half-wave-uniform (an LDS broadcast).
//   FFT         in place, decimation in frequency, radix 4 (one radix-2 stage when log2 L is odd), one barrier per stage; the result
//               stands in digit-reversed positions (pos_to_freq / freq_to_pos), which costs nothing: both passes address their
//               outputs through the map.
//   pass A      workgroup (b, t): n2 in {2b, 2b+1}, column tile t.  Loads 2 N1 row segments of 64 features (128 B of bf16: full cache
//               lines, 16 B per lane), converts to fp32, transforms along n1, multiplies by W_N^{n2 k1} (two table lookups and one
//               complex multiply) and writes the intermediate as [tile][k1][n2][32 columns]: 512 contiguous bytes per k1.
//   pass B      workgroup (u, t): residues k1 = u and N1 - u.  Reads two contiguous N2 x 256 B blocks, transforms along n2, then every
//               wave scans idx with a ballot and serves its matches two at a time (one per half-wave, lanes along the columns).
//   traffic     M once + 2 x rows x features x 4 B of intermediate + the p sampled rows: 16384 x 768 bf16, p = 3276: 25 + 2 x 50 + 5 MB.
// Roofline class: HBM / Infinity Cache bandwidth (5 N log2 N flops per column: 0.9 GFLOP for 16384 x 768).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>

#include "fewbit_hip.h"

#define FEWBIT_HIDDEN __attribute__((visibility("hidden")))

namespace fewbit_hip {

// shared with the core unit of fewbit_kernels.hip
FEWBIT_HIDDEN int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

namespace dct {

constexpr int kThreads = 256;
constexpr int C = 32;                       // complex columns of a tile = 64 features
constexpr int kFeatures = 2 * C;
constexpr int kSlots = kThreads / C;        // butterflies of one column in flight per stage pass

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ f32x2 cmul(f32x2 a, f32x2 b) { return f32x2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// e^{-2 pi i num / den}, den a power of two (num / den is exact in fp32)
__device__ __forceinline__ f32x2 unit(int num, int den) {
    float s, c;
    sincospif(-2.0f * static_cast<float>(num) / static_cast<float>(den), &s, &c);
    return f32x2{c, s};
}

// radix schedule of a length-L transform: radix 4 while at least two bits remain, then one radix 2
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v / 2); }
// position P (after the in-place DIF stages) -> frequency k.  Stage i with radix r_i on blocks of length L_i leaves digit q_i
// (k = q_1 + r_1 q_2 + r_1 r_2 q_3 + ...) in sub-block q_i: P = sum q_i L_i / r_i.
template <int L> __host__ __device__ __forceinline__ int pos_to_freq(int p) {
    int k = 0, mult = 1, len = L;
#pragma unroll
    for (int bits = ilog2(L); bits > 0;) {
        const int r = bits >= 2 ? 4 : 2, s = len / r, q = p / s;
        p -= q * s;
        k += q * mult;
        mult *= r;
        len = s;
        bits -= bits >= 2 ? 2 : 1;
    }
    return k;
}
template <int L> __host__ __device__ __forceinline__ int freq_to_pos(int k) {
    int p = 0, len = L;
#pragma unroll
    for (int bits = ilog2(L); bits > 0;) {
        const int r = bits >= 2 ? 4 : 2, s = len / r, q = k % r;
        k /= r;
        p += q * s;
        len = s;
        bits -= bits >= 2 ? 2 : 1;
    }
    return p;
}

// One stage of the in-place transform of the tile [2][L][C] along its middle axis: blocks of length LEN, radix R, twiddles
// tw[m] = W_L^m.  Thread (c = tid % 32, slot = tid / 32) takes the butterflies slot, slot + 8, ... of column c of both rows.
template <int L, int LEN, int R> __device__ __forceinline__ void stage(f32x2 *tile, const f32x2 *tw, int c, int slot) {
    constexpr int S = LEN / R, kPerRow = L / R;
#pragma unroll
    for (int bid = slot; bid < 2 * kPerRow; bid += kSlots) {
        const int row = bid / kPerRow, b = bid % kPerRow, block = b / S, ss = b % S;
        f32x2 *p = tile + (row * L + block * LEN + ss) * C + c;
        if constexpr (R == 4) {
            const f32x2 a0 = p[0], a1 = p[S * C], a2 = p[2 * S * C], a3 = p[3 * S * C];
            const f32x2 t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
            const f32x2 t3 = f32x2{d.y, -d.x};                                  // -i (a1 - a3)
            const f32x2 w1 = tw[(L / LEN) * ss], w2 = tw[(L / LEN) * ss * 2], w3 = tw[(L / LEN) * ss * 3];
            p[0] = t0 + t2;
            p[S * C] = cmul(t1 + t3, w1);
            p[2 * S * C] = cmul(t0 - t2, w2);
            p[3 * S * C] = cmul(t1 - t3, w3);
        } else {
            const f32x2 a0 = p[0], a1 = p[S * C];
            p[0] = a0 + a1;
            p[S * C] = cmul(a0 - a1, tw[(L / LEN) * ss]);
        }
    }
    __syncthreads();
}

template <int L, int LEN = L> __device__ __forceinline__ void fft_tile(f32x2 *tile, const f32x2 *tw, int c, int slot) {
    if constexpr (LEN >= 4) {
        stage<L, LEN, 4>(tile, tw, c, slot);
        fft_tile<L, LEN / 4>(tile, tw, c, slot);
    } else if constexpr (LEN == 2) {
        stage<L, LEN, 2>(tile, tw, c, slot);
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

// the piece `piece` of the 64-feature segment of row `row` that starts at feature f0, as fp32 (zeros beyond `features`)
template <int DT> __device__ __forceinline__ void load_piece(const void *x, size_t row, size_t ld, size_t f0, int piece, size_t features, float (&v)[In<DT>::kPieceFeatures]) {
    constexpr int PF = In<DT>::kPieceFeatures;
    const size_t f = f0 + static_cast<size_t>(piece) * PF;
    if constexpr (DT == FEWBIT_F32) {
        const float *p = static_cast<const float *>(x) + row * ld + f;
        if (f + PF <= features) {
            typedef f32x4 __attribute__((aligned(4))) f32x4u;
            const f32x4 q = *reinterpret_cast<const f32x4u *>(p);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = q[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = f + e < features ? p[e] : 0.0f;
        }
    } else {
        const uint16_t *p = static_cast<const uint16_t *>(x) + row * ld + f;
        if (f + PF <= features) {
            typedef u32x4 __attribute__((aligned(2))) u32x4u;
            const u32x4 q = *reinterpret_cast<const u32x4u *>(p);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] = half_to_float(q[e] & 0xffffu, DT);
                v[2 * e + 1] = half_to_float(q[e] >> 16, DT);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f + e < features ? half_to_float(p[e], DT) : 0.0f;
        }
    }
}

constexpr int kFine = 128;                  // W_N^e = fine[e % 128] * coarse[e / 128]

// ---- pass A -----------------------------------------------------------------------------------------------------------------
// grid (N2 / 2, column tiles).  inter: [tile][k1][n2][C] complex fp32.
template <int DT, int N1, int N2>
__global__ __launch_bounds__(kThreads, 2) void dct_pass_a_kernel(const void *__restrict__ x, size_t features, size_t ld, f32x2 *__restrict__ inter) {
    constexpr int N = N1 * N2, kCoarse = N / kFine > 0 ? N / kFine : 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    f32x2 *tile = reinterpret_cast<f32x2 *>(lds_raw);                 // [2][N1][C]
    f32x2 *tw = tile + 2 * N1 * C;                                    // W_N1^m, m < N1
    f32x2 *fine = tw + N1, *coarse = fine + kFine;                    // W_N^m (m < 128), W_N^{128 m}
    const int tid = threadIdx.x, c = tid % C, slot = tid / C;
    const int b = blockIdx.x;
    const size_t t = blockIdx.y, f0 = t * kFeatures;

    // ---- loads first (all in flight), tables while they travel
    constexpr int PF = In<DT>::kPieceFeatures, PPS = In<DT>::kPiecesPerSegment, kPieces = 2 * N1 * PPS / kThreads;
    static_assert(2 * N1 * PPS % kThreads == 0, "whole pieces per thread");
    float v[kPieces][PF];
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int pid = tid + kThreads * i, seg = pid / PPS, piece = pid % PPS, n1 = seg >> 1, r = seg & 1;
        const int n = N2 * n1 + 2 * b + r;                              // index into the reordered sequence v
        const size_t row = n < N / 2 ? 2 * static_cast<size_t>(n) : 2 * static_cast<size_t>(N - 1 - n) + 1;
        load_piece<DT>(x, row, ld, f0, piece, features, v[i]);
    }
    for (int m = tid; m < N1; m += kThreads) tw[m] = unit(m, N1);
    for (int m = tid; m < kFine; m += kThreads) fine[m] = unit(m, N);
    for (int m = tid; m < kCoarse; m += kThreads) coarse[m] = unit(m * kFine, N);
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int pid = tid + kThreads * i, seg = pid / PPS, piece = pid % PPS, n1 = seg >> 1, r = seg & 1;
        f32x4 *dst = reinterpret_cast<f32x4 *>(tile + (r * N1 + n1) * C + piece * (PF / 2));
#pragma unroll
        for (int e = 0; e < PF / 4; ++e) dst[e] = f32x4{v[i][4 * e], v[i][4 * e + 1], v[i][4 * e + 2], v[i][4 * e + 3]};
    }
    __syncthreads();

    fft_tile<N1>(tile, tw, c, slot);

    // ---- twiddle + store: unit = two complex columns (16 B) of one (P, r); 32 consecutive lanes = the 512 contiguous bytes of one k1
    constexpr int kUnits = 2 * N1 * (C / 2) / kThreads;
#pragma unroll 4
    for (int i = 0; i < kUnits; ++i) {
        const int uid = tid + kThreads * i, c2 = uid % (C / 2), r = (uid / (C / 2)) & 1, p = uid / C;
        const int k1 = pos_to_freq<N1>(p), e = (2 * b + r) * k1;
        const f32x2 w = kCoarse > 1 ? cmul(fine[e % kFine], coarse[e / kFine]) : fine[e % kFine];
        const f32x4 z = *reinterpret_cast<const f32x4 *>(tile + (r * N1 + p) * C + 2 * c2);
        const f32x2 a = cmul(f32x2{z[0], z[1]}, w), bb = cmul(f32x2{z[2], z[3]}, w);
        f32x4 *dst = reinterpret_cast<f32x4 *>(inter + ((t * N1 + k1) * N2 + 2 * b + r) * C + 2 * c2);
        *dst = f32x4{a.x, a.y, bb.x, bb.y};
    }
}

// ---- pass B -----------------------------------------------------------------------------------------------------------------
// grid (N1 / 2 + 1, column tiles): residues k1 = u and (N1 - u) % N1.
template <int DT, int N1, int N2>
__global__ __launch_bounds__(kThreads, 2) void dct_pass_b_kernel(const f32x2 *__restrict__ inter, const int64_t *__restrict__ idx, size_t proj, size_t features,
                                                                  float scale, void *__restrict__ out) {
    constexpr int N = N1 * N2;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    f32x2 *tile = reinterpret_cast<f32x2 *>(lds_raw);                 // [2][N2][C]
    f32x2 *tw = tile + 2 * N2 * C;                                    // W_N2^m
    const int tid = threadIdx.x, c = tid % C, slot = tid / C, lane = tid & 63, half = lane >> 5;
    const int u = blockIdx.x, k1a = u, k1b = (N1 - u) % N1;
    const size_t t = blockIdx.y, f0 = t * kFeatures;

    constexpr int kPieces = 2 * N2 * (C / 2) / kThreads;              // 16-byte pieces (two complex) per thread
    f32x4 v[kPieces];
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int pid = tid + kThreads * i, r = pid / (N2 * (C / 2)), rest = pid % (N2 * (C / 2));
        const f32x2 *src = inter + (t * N1 + (r == 0 ? k1a : k1b)) * static_cast<size_t>(N2) * C;
        v[i] = reinterpret_cast<const f32x4 *>(src)[rest];
    }
    for (int m = tid; m < N2; m += kThreads) tw[m] = unit(m, N2);
#pragma unroll
    for (int i = 0; i < kPieces; ++i) reinterpret_cast<f32x4 *>(tile)[tid + kThreads * i] = v[i];
    __syncthreads();

    fft_tile<N2>(tile, tw, c, slot);

    // ---- the sampled rows of this workgroup's two residue classes.  Every wave scans its share of idx; matches are served two
    // at a time, one per half-wave, lanes along the 32 complex columns.
    const float base = scale * __builtin_sqrtf(0.5f / static_cast<float>(N));          // ortho: sqrt(1 / 2N) (k > 0), sqrt(1 / 4N) (k = 0)
    for (size_t i0 = 0; i0 < proj; i0 += kThreads) {
        const size_t i = i0 + tid;
        int k = -1;
        if (i < proj) {
            const int kk = static_cast<int>(idx[i]) & (N - 1);
            const int k1 = kk % N1;
            if (k1 == k1a || k1 == k1b) k = kk;
        }
        unsigned long long mask = __ballot(k >= 0);
        while (mask != 0) {                                             // wave-uniform
            const int l0 = __builtin_ctzll(mask);
            mask &= mask - 1;
            int l1 = l0;
            if (mask != 0) {
                l1 = __builtin_ctzll(mask);
                mask &= mask - 1;
            }
            const int src = half ? l1 : l0;
            const int km = __shfl(k, src);                              // (every lane takes part in the exchange, then an odd match out idles the upper half)
            if (half == 1 && l1 == l0) continue;
            const size_t j = i0 + (tid & ~63) + src;                    // the sample this half-wave serves
            const int k1 = km % N1, k2 = km / N1;
            const int r = k1 == k1a ? 0 : 1;
            const int k2m = k1 == 0 ? (N2 - k2) % N2 : N2 - 1 - k2;     // N - k = (N1 - k1) + N1 k2m
            const f32x2 zk = tile[(r * N2 + freq_to_pos<N2>(k2)) * C + c];
            f32x2 zm = tile[((1 - r) * N2 + freq_to_pos<N2>(k2m)) * C + c];
            zm.y = -zm.y;                                               // conj Z[N - k]
            const f32x2 va = (zk + zm) * 0.5f, d = (zk - zm) * 0.5f, vb = f32x2{d.y, -d.x};
            float sn, cs;
            sincospif(static_cast<float>(km) / static_cast<float>(2 * N), &sn, &cs);
            const float f = (km == 0 ? 0.70710678118654752f : 1.0f) * 2.0f * base;
            const float ya = (cs * va.x + sn * va.y) * f, yb = (cs * vb.x + sn * vb.y) * f;     // Re(e^{-i theta} V)
            const size_t fa = f0 + 2 * c;
            if constexpr (DT == FEWBIT_F32) {
                float *o = static_cast<float *>(out) + j * features + fa;
                if (fa + 1 < features) *reinterpret_cast<f32x2 *>(o) = f32x2{ya, yb};
                else if (fa < features) o[0] = ya;
            } else {
                uint16_t *o = static_cast<uint16_t *>(out) + j * features + fa;
                uint16_t ha, hb;
                if constexpr (DT == FEWBIT_BF16) {
                    ha = __builtin_bit_cast(uint16_t, static_cast<__bf16>(ya));
                    hb = __builtin_bit_cast(uint16_t, static_cast<__bf16>(yb));
                } else {
                    ha = __builtin_bit_cast(uint16_t, static_cast<_Float16>(ya));
                    hb = __builtin_bit_cast(uint16_t, static_cast<_Float16>(yb));
                }
                if (fa + 1 < features) {
                    typedef uint32_t __attribute__((aligned(2))) u32u;
                    *reinterpret_cast<u32u *>(o) = static_cast<uint32_t>(ha) | (static_cast<uint32_t>(hb) << 16);
                } else if (fa < features) {
                    o[0] = ha;
                }
            }
        }
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------------
struct Split { int n1, n2; };
// rows = N1 x N2 with 16 <= N2 <= N1 <= 128: 256 .. 16384 rows
bool split_rows(size_t rows, Split &s) {
    if (rows < 256 || rows > 16384 || (rows & (rows - 1)) != 0) return false;
    int bits = 0;
    while ((static_cast<size_t>(1) << bits) < rows) ++bits;
    s.n1 = 1 << ((bits + 1) / 2);
    s.n2 = 1 << (bits / 2);
    return true;
}
size_t tiles_of(size_t features) { return (features + kFeatures - 1) / kFeatures; }
size_t inter_bytes(size_t rows, size_t features) { return tiles_of(features) * rows * C * sizeof(f32x2); }

template <int L> constexpr size_t lds_bytes_a(size_t n) { return (2 * L * C + L + kFine + (n / kFine > 0 ? n / kFine : 1)) * sizeof(f32x2); }
template <int L> constexpr size_t lds_bytes_b() { return (2 * L * C + L) * sizeof(f32x2); }

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

template <int DT, int N1, int N2>
int launch(const void *m, size_t features, size_t ld, const int64_t *idx, size_t proj, float scale, void *out, f32x2 *inter, hipStream_t s) {
    static std::atomic<unsigned long long> done_a{0}, done_b{0};
    constexpr size_t la = lds_bytes_a<N1>(static_cast<size_t>(N1) * N2), lb = lds_bytes_b<N2>();
    if (const int rc = opt_in(dct_pass_a_kernel<DT, N1, N2>, la, done_a)) return rc;
    if (const int rc = opt_in(dct_pass_b_kernel<DT, N1, N2>, lb, done_b)) return rc;
    const unsigned tiles = static_cast<unsigned>(tiles_of(features));
    hipLaunchKernelGGL((dct_pass_a_kernel<DT, N1, N2>), dim3(N2 / 2, tiles), dim3(kThreads), la, s, m, features, ld, inter);
    hipLaunchKernelGGL((dct_pass_b_kernel<DT, N1, N2>), dim3(N1 / 2 + 1, tiles), dim3(kThreads), lb, s, inter, idx, proj, features, scale, out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FEWBIT_ERR_LAUNCH, "sampled_dct: %s", hipGetErrorString(e));
    return FEWBIT_OK;
}

template <int DT>
int launch_rows(Split sp, const void *m, size_t features, size_t ld, const int64_t *idx, size_t proj, float scale, void *out, f32x2 *inter, hipStream_t s) {
#define FB_DCT_CASE(A, B) \
    if (sp.n1 == A && sp.n2 == B) return launch<DT, A, B>(m, features, ld, idx, proj, scale, out, inter, s);
    FB_DCT_CASE(16, 16) FB_DCT_CASE(32, 16) FB_DCT_CASE(32, 32) FB_DCT_CASE(64, 32) FB_DCT_CASE(64, 64) FB_DCT_CASE(128, 64) FB_DCT_CASE(128, 128)
#undef FB_DCT_CASE
    return fail(FEWBIT_ERR_UNSUPPORTED, "sampled_dct: no kernel for %d x %d rows", sp.n1, sp.n2);
}

}  // namespace dct
}  // namespace fewbit_hip

using namespace fewbit_hip;
using namespace fewbit_hip::dct;

extern "C" {

size_t fewbit_hip_sampled_dct_workspace(int dtype, size_t rows, size_t features, size_t proj) {
    Split sp;
    (void)dtype;
    if (features == 0 || proj == 0 || !split_rows(rows, sp)) return 0;
    return inter_bytes(rows, features);
}

int fewbit_hip_sampled_dct(int dtype, const void *m, size_t rows, size_t features, size_t ld, const int64_t *idx, size_t proj, double scale, void *out,
                           void *workspace, size_t workspace_bytes, void *stream) {
    if (dtype != FEWBIT_F32 && dtype != FEWBIT_F16 && dtype != FEWBIT_BF16) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_dct: unknown dtype %d", dtype);
    if (proj == 0 || features == 0) return FEWBIT_OK;
    Split sp;
    if (!split_rows(rows, sp)) return fail(FEWBIT_ERR_UNSUPPORTED, "sampled_dct: rows = %zu is not a power of two in [256, 16384]", rows);
    if (m == nullptr || idx == nullptr || out == nullptr) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_dct: null pointer");
    if (ld < features) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_dct: leading dimension %zu < features %zu", ld, features);
    const size_t need = inter_bytes(rows, features);
    if (workspace == nullptr || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
        return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sampled_dct: a 16-byte aligned workspace of %zu bytes is needed (fewbit_hip_sampled_dct_workspace), got %zu", need, workspace_bytes);
    if (tiles_of(features) > 65535) return fail(FEWBIT_ERR_UNSUPPORTED, "sampled_dct: more than 65535 column tiles");
    hipStream_t s = static_cast<hipStream_t>(stream);
    f32x2 *inter = static_cast<f32x2 *>(workspace);
    const float fs = static_cast<float>(scale);
    switch (dtype) {
    case FEWBIT_F32: return launch_rows<FEWBIT_F32>(sp, m, features, ld, idx, proj, fs, out, inter, s);
    case FEWBIT_F16: return launch_rows<FEWBIT_F16>(sp, m, features, ld, idx, proj, fs, out, inter, s);
    default: return launch_rows<FEWBIT_BF16>(sp, m, features, ld, idx, proj, fs, out, inter, s);
    }
}

}  // extern "C"
