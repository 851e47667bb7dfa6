// fewbit_sketch.hip -- the random-projection products of the randomized linear layers on gfx950:
//
//     P = scale * S . M        S: proj x rows, random (Rademacher +-1 or Gaussian N(0,1)), a FUNCTION of a 64-bit seed
//                              M: rows x features (the layer's input X or the incoming gradient G), bf16 / fp16 / fp32
//                              P: proj x features, the dtype of M
//
// What it replaces in the reference (skolai/fewbit): `proj = T.randn((proj_features, rows))` / `T.randint(...) - 0.5`
// followed by `proj @ input_view` in LinearGRPFunc.forward (fewbit/functional/linear.py:133-146) and the same pair in
// .backward (:195-208).  There S is a tensor in device memory (proj x rows fp32 elements, drawn twice:
// the backward re-draws it from the saved generator state) and the product is a library GEMM.  Here S is a function of
// (seed, row, column), defined in the A-operand layout of v_mfma_f32_32x32x16_{bf16,f16}; forward and backward regenerate the
// same S from the seed alone.  Two data paths (sketch_entry picks one; both give the same S):
//   fused         every lane of the matrix pipe computes its own A fragment in registers and feeds it straight to the MFMA: S costs
//                 no HBM, L2 or LDS byte.  Rademacher always (8 VALU per fragment); Gaussian for layers no wider than one tile.
//   from memory   (Gaussian, more than one column tile) a VALU-only kernel writes S ONCE, as ready-made A fragments, into the
//                 workspace and the product kernel reads them back: a Gaussian fragment costs as many issue cycles as the 8 MFMAs
//                 it feeds, and on a CDNA4 SIMD nothing hides VALU work behind the matrix pipe (scratch/gen_bench.hip,
//                 profiles/r05_gen_bench.txt) -- regenerating S per column tile was the kernel's bound.
//
// ---- the definition of S (a pure function; tests/sketch_reference.py evaluates the same formulas with numpy) -------------
//   philox(c0, c1, c2, c3) = Philox4x32-10 with key (seed_lo, seed_hi)          [Salmon et al. 2011; the generator behind
//                                                                               torch.cuda's and cuRAND's default engine]
//   Rademacher:  S[i][r] = bit ? -1 : +1,   bit = bit ((j%2 ? 31 : 15) - (4*(s%4) + j/2)) of word s/4 of philox(i, 2*(r/256) + h, 0, 0)
//                with s = (r%256)/16, h = (r/8)%2, j = r%8        (one call = 128 signs = this lane's 16 MFMA steps; the bit
//                order is the one that turns a word into operand sign bits with one shift and one and-or per dword)
//   Gaussian:    S[i][r] = Box-Muller on 16-bit uniforms of a 32-bit word w: u1 = ((w & 0xffff) + 0.5) / 65536, u2 = (w >> 16) / 65536,
//                rad = sqrt(-2 ln u1), S = rad * cos(2 pi u2) for even j, rad * sin(2 pi u2) for odd j, rounded to the operand dtype
//                (ln/sqrt/sin/cos are the hardware's v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32, ~1 ulp each -- the host model
//                uses libm, and the test allows one step of the 16-bit operand dtype).  The word: with s, h, j as above and
//                q = j/2, w = output number 2 s + q%2 of the xoshiro128++ stream [Blackman & Vigna 2019] whose state is
//                philox(i, 2*(r/256) + h, q/2, 2) -- per lane and 256-row block TWO streams, each seeded by one Philox call and
//                advanced two words per MFMA step; stream t feeds operand dwords 2t, 2t+1 of every fragment (so that the two
//                waves of the 128 x 512 tile, which share their fragments, can each run one).
//
// ---- tiling (three shapes, `Tile<W, NH>` below; the host picks one per call, make_plan) -------------------------------------
//   workgroup  W waves (4 or 8); wave w owns 32 rows of S x 256 features = 8 MFMA column blocks of 32 -> 8 x 16 fp32
//              accumulators per lane.  Waves split the S rows (and, in the 128 x 512 tile, the columns in two halves whose wave
//              pairs share their A fragments through LDS), so no S element is generated twice inside a workgroup.
//   K loop     over the rows of M in stages of 64 or 128, double-buffered in LDS, two stages deep: while stage s is multiplied
//              stage s+1 (loaded during stage s-1) is transposed into the other buffer and the loads of stage s+2 go out; the
//              staging work and the loads are pinned into the MFMA stream slot by slot; one barrier per stage.
//   M -> LDS   M is row-major (features contiguous) but the MFMA B operand wants 8 consecutive K values (rows) of ONE
//              feature per lane.  Each thread loads an 8-row x 8-feature block (8 x 16 B, coalesced along the rows),
//              transposes it in registers with 32 v_perm_b32 and writes 8 x 16 B: the LDS image is [octet of rows][chunk]
//              with chunk = 32*(feature%8) + feature/8 -- chunk-major, so that the 8 ds_write_b128 of a thread are
//              conflict-free AND a B fragment is ONE conflict-free ds_read_b128.  The price is a permuted column order
//              inside the tile: MFMA block t, column c  <->  feature 8c + t.  It costs nothing: the epilogue finds the 8
//              accumulators of a lane (t = 0..7) holding 8 CONSECUTIVE features -> one 16/32-byte store per output row.
//   split K    gridDim.z slices of the rows (multiples of 256) when the tile grid alone cannot fill 256 CUs (proj x
//              features is small, rows is long): slices write partial sums (fp32; bf16 when the result is bf16), a second kernel
//              adds them IN A FIXED ORDER (deterministic: the same seed gives the same bits), scales and casts.
// Roofline class: MFMA (bf16 dense peak 2.5 PFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md); flops = 2*proj*rows*features.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "fewbit_hip.h"
#include "fewbit_philox.h"

#define FEWBIT_HIDDEN __attribute__((visibility("hidden")))

namespace fewbit_hip {

// shared with the core unit of fewbit_kernels.hip
extern FEWBIT_HIDDEN thread_local char g_last_error[256];
FEWBIT_HIDDEN int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

namespace sketch {

constexpr int BN = 256;                     // features per tile
constexpr int NT = BN / 32;                 // MFMA column blocks per wave
// A workgroup of W waves owns 32*W rows of S and multiplies them with K stages of 16*W rows of M (one 8x8 block per thread):
//   W = 4:  128 x 256 tile, stages of  64 rows, 2 x 32 KiB of LDS, two workgroups per CU
//   W = 8:  256 x 256 tile, stages of 128 rows, 2 x 64 KiB of LDS, one workgroup per CU (half the staging work, LDS writes and
//           L2 reads per MFMA; the upper four waves do their staging four steps later than the lower four)
//   W = 8, NH = 2 (the Gaussian sketch): 128 x 512 tile -- the eight waves are 4 row groups x 2 column halves, waves (g, 0) and
//           (g, 1) need the SAME rows of S, so each generates half of the stage's A fragments and hands them to the other
//           through LDS (16 KiB per stage): every element of S is generated once per 512 columns instead of once per 256, which
//           halves the generator work per MFMA -- the Gaussian sketch is bound by instruction issue (EXPERIMENTS.md 3.1).
//           Stages of 64 rows, 2 x 64 KiB (M) + 2 x 16 KiB (A fragments) = all 160 KiB of the CU.
template <int W, int NH = 1> struct Tile {
    static constexpr int kThreads = 64 * W, RG = W / NH, BM = 32 * RG, BNT = BN * NH, BK = 16 * W / NH, kSteps = BK / 16;
    static constexpr int kStageBytes = BK * BNT * 2;
    static constexpr int kABytes = NH > 1 ? RG * kSteps * 1024 : 0;     // one stage's A fragments: 1 KiB per (row group, step)
    static constexpr int kLdsBytes = 2 * kStageBytes + 2 * kABytes;
    static constexpr int kStagesPerBlock = 256 / BK;           // stages per 256-row Rademacher block
    static constexpr int kChunksPerRow = BNT / 8;              // 16-byte chunks per octet of rows in the LDS image
};
// internal "distribution" of the product kernel: the A fragments were written to memory beforehand (sketch_fragments_kernel)
constexpr int kFromMemory = 2;
constexpr int kFragAhead = 4;               // MFMA steps a fragment load runs ahead of its use (registers: 4 dwords per step)
constexpr int kGaussianRounds = 10;         // Philox rounds of the calls that seed the Gaussian streams

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// (Philox4x32-10, `Key`: fewbit_philox.h, shared with the sampled cosine transform)

// ---- xoshiro128++ 1.0 (Blackman & Vigna, "Scrambled linear pseudorandom number generators", 2019; public-domain reference
// xoshiro128plusplus.c) -- 9 one-cycle VALU instructions per 32 bits, against ~85 issue slots for the 128 bits of a Philox call
__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
__host__ __device__ __forceinline__ uint32_t xoshiro128pp(uint32_t (&s)[4]) {
    const uint32_t result = rotl32(s[0] + s[3], 7) + s[0];
    const uint32_t t = s[1] << 9;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl32(s[3], 11);
    return result;
}

// operand type of the matrix pipe: bf16 for bf16 AND fp32 inputs (fp32 is rounded to bf16 while it is staged), fp16 for fp16
template <int DT> struct Operand {
    typedef bf16x8 frag;
    static constexpr uint32_t kOnes = 0x3F803F80u;        // (+1.0, +1.0)
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        bf16x2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
        return __builtin_bit_cast(uint32_t, v);
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Operand<FEWBIT_F16> {
    typedef f16x8 frag;
    static constexpr uint32_t kOnes = 0x3C003C00u;
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
        f16x2 v = {static_cast<_Float16>(a), static_cast<_Float16>(b)};
        return __builtin_bit_cast(uint32_t, v);
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// ---- S as a function -----------------------------------------------------------------------------------------------------
// Rademacher: the 8 signs of MFMA step s (0..15) of the 256-row block whose 128 bits are `w`, as 4 packed operand dwords:
// dword q = (+-1, +-1) with the sign bits taken from bits 15 - sh and 31 - sh of word s/4, sh = 4*(s%4) + q
template <int DT> __device__ __forceinline__ u32x4 rademacher_fragment(const uint32_t (&w)[4], int s) {
    const uint32_t word = w[s >> 2];
    u32x4 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = ((word << (4 * (s & 3) + q)) & 0x80008000u) | Operand<DT>::kOnes;
    return a;
}

// Gaussian: one 32-bit word -> one Box-Muller pair (u1 = (x + 0.5) / 65536 is ONE exact v_fma_f32)
__device__ __forceinline__ void box_muller(uint32_t w, float &z0, float &z1) {
    const float u1 = __builtin_fmaf(static_cast<float>(w & 0xffffu), 1.0f / 65536.0f, 0.5f / 65536.0f);
    const float u2 = static_cast<float>(w >> 16) * (1.0f / 65536.0f);
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));      // -2 ln u = -2 ln2 * log2 u
    z0 = rad * __builtin_amdgcn_cosf(u2);                                                             // v_cos_f32 takes turns
    z1 = rad * __builtin_amdgcn_sinf(u2);
}

// the same in two halves (for the woven generator: first half behind one MFMA, second half behind the next)
__device__ __forceinline__ void box_muller_begin(uint32_t w, float &l2, float &u2) {
    l2 = __builtin_amdgcn_logf(__builtin_fmaf(static_cast<float>(w & 0xffffu), 1.0f / 65536.0f, 0.5f / 65536.0f));
    u2 = static_cast<float>(w >> 16) * (1.0f / 65536.0f);
}
template <int DT> __device__ __forceinline__ uint32_t box_muller_end(float l2, float u2) {
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * l2);
    return Operand<DT>::pack(rad * __builtin_amdgcn_cosf(u2), rad * __builtin_amdgcn_sinf(u2));
}

// one 32-bit word -> two normals, packed as one operand dword (even element low)
template <int DT> __device__ __forceinline__ uint32_t gaussian_pair(uint32_t w) {
    float z0, z1;
    box_muller(w, z0, z1);
    return Operand<DT>::pack(z0, z1);
}

// ---- M -> registers -> LDS -----------------------------------------------------------------------------------------------
// A thread stages one 8-row x 8-feature block per stage: 8 pieces of 8 consecutive features, addressed as a wave-uniform
// 64-bit stage base plus a 32-bit per-lane byte offset per row (fixed for the whole K loop: no per-stage address arithmetic).
//   * feature chunks outside the matrix (the tile overhangs it) are read from column 0 instead: those columns of the tile
//     only feed outputs that are never stored, so their values do not matter -- the address just has to be valid;
//   * rows outside the slice exist only in its LAST stage: that stage is fetched by the guarded variant (row clamped to a
//     valid one, data replaced by zeros before it is staged); every other stage runs with no compare and no select;
//   * features % 8 != 0 (RAGGED): element-wise guarded loads in every stage (slow; a layer width is a multiple of 8).
// The loads of a stage are issued back to back at its top and first touched AFTER its MFMAs (a load under a lane-divergent
// branch, or a select right behind it, would make the compiler wait for the data on the spot).
template <int DT> struct RawPiece { u32x4 v; };
template <> struct RawPiece<FEWBIT_F32> { f32x4 lo, hi; };
template <int DT> struct RawBlock { RawPiece<DT> row[8]; };
struct Block8x8 { u32x4 row[8]; };

template <int DT> __device__ __forceinline__ constexpr int elem_size() { return DT == FEWBIT_F32 ? 4 : 2; }

template <int DT> __device__ __forceinline__ RawPiece<DT> load_raw(const uint8_t *p) {
    RawPiece<DT> r;
    if constexpr (DT == FEWBIT_F32) {
        typedef f32x4 __attribute__((aligned(4))) f32x4u;
        r.lo = *reinterpret_cast<const f32x4u *>(p);
        r.hi = *reinterpret_cast<const f32x4u *>(p + 16);
    } else {
        typedef u32x4 __attribute__((aligned(2))) u32x4u;
        r.v = *reinterpret_cast<const u32x4u *>(p);
    }
    return r;
}

// interior stage: `base` = first row of the stage, `row_bytes` = one row of M (both wave-uniform: the row advance stays in
// scalar registers and the loads take the scalar-base form), off0 = byte offset of this thread's piece of the stage's row 0
template <int DT> __device__ __forceinline__ void fetch_fast(RawBlock<DT> &b, const uint8_t *base, size_t row_bytes, uint32_t off0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) b.row[j] = load_raw<DT>(base + j * row_bytes + off0);
}

// last stage of a slice: rows >= rows_left are clamped to the last valid one (and zeroed by finish_block)
template <int DT> __device__ __forceinline__ void fetch_clamped(RawBlock<DT> &b, const uint8_t *base, size_t row_bytes, uint32_t off0, int row0, int rows_left) {
    const uint32_t rb = static_cast<uint32_t>(row_bytes);      // (a stage spans less than 2 GiB: checked by the host)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int last = rows_left - 1 - row0;                 // index (within this thread's octet) of the last valid row; may be < 0
        const int jc = j <= last ? j : (last > 0 ? last : 0);
        const uint32_t o = last >= 0 ? off0 + static_cast<uint32_t>(jc) * rb : off0 - static_cast<uint32_t>(row0) * rb;
        b.row[j] = load_raw<DT>(base + o);
    }
}

// element-wise guarded (ragged feature count, or anything else the vector paths cannot take)
template <int DT>
__device__ __forceinline__ void fetch_guarded(RawBlock<DT> &b, const void *m, size_t ld, size_t row0, size_t row_end, size_t f0, size_t features) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const size_t row = row0 + j;
        if constexpr (DT == FEWBIT_F32) {
            const float *p = static_cast<const float *>(m) + row * ld + f0;
            float v[8];
            for (int e = 0; e < 8; ++e) v[e] = (row < row_end && f0 + e < features) ? p[e] : 0.0f;
            b.row[j].lo = f32x4{v[0], v[1], v[2], v[3]};
            b.row[j].hi = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            const uint16_t *p = static_cast<const uint16_t *>(m) + row * ld + f0;
            u32x4 z = {0u, 0u, 0u, 0u};
            for (int e = 0; e < 8; ++e) {
                const uint32_t hw = (row < row_end && f0 + e < features) ? p[e] : 0u;
                z[e >> 1] |= hw << (16 * (e & 1));
            }
            b.row[j].v = z;
        }
    }
}

// raw pieces -> operand dwords (fp32 is rounded to bf16 here); MASK: rows at or beyond `valid` (within the octet) -> 0
template <int DT, bool MASK> __device__ __forceinline__ void finish_block(const RawBlock<DT> &r, Block8x8 &b, int valid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        u32x4 z;
        if constexpr (DT == FEWBIT_F32) {
            z[0] = Operand<DT>::pack(r.row[j].lo[0], r.row[j].lo[1]); z[1] = Operand<DT>::pack(r.row[j].lo[2], r.row[j].lo[3]);
            z[2] = Operand<DT>::pack(r.row[j].hi[0], r.row[j].hi[1]); z[3] = Operand<DT>::pack(r.row[j].hi[2], r.row[j].hi[3]);
        } else {
            z = r.row[j].v;
        }
        if constexpr (MASK) {
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = j < valid ? z[e] : 0u;
        }
        b.row[j] = z;
    }
}

// transpose the 8x8 block and store it: feature w of the block -> chunk 32*w + fc of octet `octet` of the stage buffer
// (`pitch` = features of the tile: 256, or 512 with two column halves -- feature chunk fc = 32*half + c' then goes to chunk
// 256*half + 32*w + c' of its octet)
template <int PITCH> __device__ __forceinline__ void store_feature(const Block8x8 &b, uint8_t *stage, int octet, int fc, int w) {
    const int e = w >> 1;
    const uint32_t sel = (w & 1) ? 0x07060302u : 0x05040100u;
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = __builtin_amdgcn_perm(b.row[2 * q + 1][e], b.row[2 * q][e], sel);
    const int chunk = PITCH == BN ? 32 * w + fc : 256 * (fc >> 5) + 32 * w + (fc & 31);
    *reinterpret_cast<u32x4 *>(stage + (static_cast<size_t>(octet) * PITCH + chunk) * 16) = o;
}

template <int PITCH> __device__ __forceinline__ void store_block(const Block8x8 &b, uint8_t *stage, int octet, int fc) {
#pragma unroll
    for (int w = 0; w < 8; ++w) store_feature<PITCH>(b, stage, octet, fc, w);
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------
// grid: x = column tiles (256 features), y = row tiles of S (128), z = K slices.  PARTIAL (1: fp32, 2: bf16): write partial sums to
// `out` + z * proj * features (no scale); otherwise the scaled result in the dtype of M.
template <int DIST, int DT, int PARTIAL, bool RAGGED, int W, int NH>
__global__ __launch_bounds__(64 * W, 2) void sketch_kernel(const void *__restrict__ m, size_t rows, size_t features, size_t ld, size_t proj,
                                                           Key key, const Key *__restrict__ key_dev, float scale, void *__restrict__ out,
                                                           size_t kslice, const void *__restrict__ frags, size_t frag_steps) {
    typedef Tile<W, NH> T_;
    static_assert(DIST != kFromMemory || NH == 1, "fragments from memory: the one-half tiles only");
    if (key_dev != nullptr) key = *key_dev;          // the seed of a call inside a hipGraph lives in device memory (one scalar load)
    constexpr int BM = T_::BM, BK = T_::BK, BNT = T_::BNT, RG = T_::RG, kStageBytes = T_::kStageBytes, kSteps = T_::kSteps;
    constexpr int kPerBlock = T_::kStagesPerBlock, kABytes = T_::kABytes;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];        // 2 * kStageBytes of M, then 2 * kABytes of A fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int rg = NH > 1 ? wave % RG : wave, hf = NH > 1 ? wave / RG : 0;    // row group of S, column half of the tile
    // Which tile this workgroup takes.  The dispatcher is observed to put workgroup b (x fastest) on XCD b % 8, each XCD with its own
    // 4 MiB L2 (MI355X_MICROARCH.md): in the grid's own order the 32 workgroups an XCD runs at a time are scattered over all column
    // tiles and some 20 row tiles, and every one of them pulls its stage of M (and its fragments of S) into that L2 separately.
    // Here XCD i walks a contiguous range of a tile list ordered slice-major, then in groups of 8 row tiles x all column tiles,
    // row tile fastest: its 32 concurrent workgroups cover ~4 column tiles x 8 row tiles of ONE slice, (4 + 8) instead of ~(12 + 20)
    // operand streams per stage.  A bijection of the grid (T1 in cdna_hip_programming.md); speed only, placement is no contract.
    // Used by the kernel that reads S from memory (two operand streams per workgroup: 16384 x 3072, p = 3276: 330.5 -> 324.9 us,
    // 768 wide 115.1 -> 109.6); the kernels that generate S themselves measured 1.5-3 % SLOWER with it (Rademacher 280.7 ->
    // 285.5 / 82.9 -> 84.4 us) and keep the grid's own order (profiles/r05_sketch_xcd_prefetch_ab.txt).
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if constexpr (DIST == kFromMemory) {
        const unsigned gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
        const unsigned orig = bx + gx * (by + gy * bz), xcd = orig % 8, q = nwg / 8, r = nwg % 8;
        const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
        constexpr unsigned GY = 8;
        const unsigned t = v % (gx * gy), grp = t / (GY * gx), first = grp * GY, gh = gy - first < GY ? gy - first : GY, u = t - grp * GY * gx;
        bz = v / (gx * gy);
        by = first + u % gh;
        bx = u / gh;
    }
    const size_t n0 = static_cast<size_t>(bx) * BNT, m0 = static_cast<size_t>(by) * BM;
    const size_t k_begin = static_cast<size_t>(bz) * kslice;
    const size_t k_end = k_begin + kslice < rows ? k_begin + kslice : rows;
    const uint32_t srow = static_cast<uint32_t>(m0 + 32 * rg + c);            // this lane's row of S

    // staging role of this thread: octet `so` (8 rows) of the stage, features [8*sfc, 8*sfc + 8) of the tile
    const int so = tid / T_::kChunksPerRow, sfc = tid % T_::kChunksPerRow;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const size_t klen = k_begin < k_end ? k_end - k_begin : 0;
    const size_t nstages = (klen + BK - 1) / BK, nfull = klen / BK;
    if (nstages == 0) return;                      // (block-uniform; cannot happen with the host's slicing)
    // this thread's piece of row j of a stage: byte offset from the stage's first row (column 0 for a chunk outside the matrix)
    constexpr int ES = elem_size<DT>();
    const size_t col = n0 + 8 * sfc < features ? n0 + 8 * sfc : 0;
    const uint32_t off0 = static_cast<uint32_t>((8 * so * ld + col) * ES);
    const size_t row_bytes = ld * ES;
    const uint8_t *stage_base = static_cast<const uint8_t *>(m) + k_begin * ld * ES;
    const size_t stage_bytes = static_cast<size_t>(BK) * ld * ES;

    RawBlock<DT> raw;
    Block8x8 blk;
    // mode of a stage's fetch: 0 interior (no guards), 1 the partial last stage (rows clamped + zeroed), 2 nothing to fetch
    auto fetch = [&](size_t st, int mode) __attribute__((always_inline)) {
        if constexpr (RAGGED) {
            if (mode != 2) fetch_guarded<DT>(raw, m, ld, k_begin + st * BK + 8 * so, k_end, n0 + 8 * sfc, features);
        } else {
            if (mode == 0) fetch_fast<DT>(raw, stage_base + st * stage_bytes, row_bytes, off0);
            else if (mode == 1) fetch_clamped<DT>(raw, stage_base + st * stage_bytes, row_bytes, off0, 8 * so, static_cast<int>(klen - st * BK));
        }
    };
    auto stage_to_lds = [&](size_t st, int mode, uint8_t *buf) __attribute__((always_inline)) {
        if (mode == 2) return;
        if (!RAGGED && mode == 1) finish_block<DT, true>(raw, blk, static_cast<int>(klen - st * BK) - 8 * so);
        else finish_block<DT, false>(raw, blk, 8);
        store_block<BNT>(blk, buf, so, sfc);
    };
    auto mode_of = [&](size_t st) { return st < nfull ? 0 : st < nstages ? 1 : 2; };

    // Software pipeline, two stages deep in registers + LDS: while stage s is multiplied (LDS buffer s%2), stage s+1 -- loaded
    // during stage s-1 -- is transposed and written to the other buffer, and the loads of stage s+2 go out into the registers
    // that frees.  Every load has a whole stage to land before it is touched; one barrier per stage.
    fetch(0, mode_of(0));
    stage_to_lds(0, mode_of(0), lds);
    fetch(1, mode_of(1));
    __syncthreads();

    uint32_t signs[4] = {0u, 0u, 0u, 0u};
    // A fragment of step `ks` of stage `st` for this lane's row of S (Rademacher: `signs` must hold the Philox words of the
    // 256-row block that contains the stage)
    // Gaussian: the xoshiro128++ streams of this lane's 256-row block -- stream t feeds operand dwords 2t and 2t + 1 of every
    // step, two words per step, in step order.  One column half (NH == 1): the wave runs both streams; two halves: wave (g, hf)
    // runs stream hf only and its partner the other one (each writes its 8 bytes of every fragment, publish_fragments)
    constexpr bool kStreams = DIST == FEWBIT_SKETCH_GAUSSIAN;
    uint32_t gs[NH > 1 ? 1 : 2][4];
    auto refresh_signs = [&](size_t st) __attribute__((always_inline)) {
        if constexpr (DIST == FEWBIT_SKETCH_RADEMACHER) {
            // one Philox call covers this lane's 16 MFMA steps = 256 rows = 4 or 2 stages (k_begin is a multiple of 256)
            if ((st & (kPerBlock - 1)) == 0)
                philox4x32(srow, static_cast<uint32_t>(2 * ((k_begin + st * BK) >> 8) + h), 0u, 0u, key, signs);
        } else if constexpr (kStreams) {
            if ((st & (kPerBlock - 1)) == 0) {
                const uint32_t blk = static_cast<uint32_t>(2 * ((k_begin + st * BK) >> 8) + h);
                if constexpr (NH > 1) {
                    philox4x32<kGaussianRounds>(srow, blk, static_cast<uint32_t>(hf), 2u, key, gs[0]);
                } else {
                    philox4x32<kGaussianRounds>(srow, blk, 0u, 2u, key, gs[0]);
                    philox4x32<kGaussianRounds>(srow, blk, 1u, 2u, key, gs[1]);
                }
            }
        }
    };
    // fragments from memory (kFromMemory): fragment (row block rb, step g) is 64 lanes x 16 bytes at ((rb * frag_steps + g) * 64
    // + lane) * 16 -- a wave walks 1 KiB per step through consecutive addresses; the load of step g + kFragAhead goes out when
    // step g's fragment is taken from its register (the buffer ends in kFragAhead steps of padding)
    const uint8_t *afrag = nullptr;
    constexpr int kAhead = kFragAhead < kSteps ? kFragAhead : kSteps;      // (at most one stage ahead)
    u32x4 apre[DIST == kFromMemory ? kAhead : 1];
    if constexpr (DIST == kFromMemory) {
        static_assert(kSteps % kAhead == 0, "the register ring is indexed by the step within a stage");
        afrag = static_cast<const uint8_t *>(frags) + (((m0 / 32 + rg) * frag_steps + (k_begin >> 4)) * 64 + lane) * 16;
#pragma unroll
        for (int d = 0; d < kAhead; ++d) apre[d] = *reinterpret_cast<const u32x4 *>(afrag + static_cast<size_t>(d) * 1024);
    }
    // (Gaussian streams and fragments from memory: called ONCE per step and in step order -- every call consumes the next words)
    auto make_fragment = [&](size_t st, int ks) __attribute__((always_inline)) -> u32x4 {
        if constexpr (DIST == kFromMemory) {
            const u32x4 a = apre[ks % kAhead];
            apre[ks % kAhead] = *reinterpret_cast<const u32x4 *>(afrag + (st * kSteps + ks + kAhead) * 1024);
            return a;
        }
        else if constexpr (DIST == FEWBIT_SKETCH_RADEMACHER) return rademacher_fragment<DT>(signs, static_cast<int>((st & (kPerBlock - 1)) * kSteps + ks));
        else {                                         // Gaussian, one column half: both streams of this lane (two halves: publish_fragments)
            u32x4 a;
            a[0] = gaussian_pair<DT>(xoshiro128pp(gs[0]));
            a[1] = gaussian_pair<DT>(xoshiro128pp(gs[0]));
            a[2] = gaussian_pair<DT>(xoshiro128pp(gs[NH > 1 ? 0 : 1]));
            a[3] = gaussian_pair<DT>(xoshiro128pp(gs[NH > 1 ? 0 : 1]));
            return a;
        }
    };
    // NH = 2: this wave's share of the A fragments of stage `st` (steps [hf * kSteps / 2, (hf + 1) * kSteps / 2)) -> the LDS
    // exchange buffer of that stage; its partner (same rows, other column half) writes the other steps
    uint8_t *abuf = lds + 2 * kStageBytes;
    auto publish_fragments = [&](size_t st, auto checked_tag) __attribute__((always_inline)) {
        if constexpr (NH > 1) {
            if constexpr (decltype(checked_tag)::value) {
                if (st >= nstages) return;             // block-uniform (the interior loop never gets here)
            }
            refresh_signs(st);
            if constexpr (kStreams) {
                // this wave's stream = dwords 2 hf, 2 hf + 1 of EVERY step of the stage (8 of the fragment's 16 bytes)
                typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
#pragma unroll
                for (int ks = 0; ks < kSteps; ++ks) {
                    u32x2 a;
                    a[0] = gaussian_pair<DT>(xoshiro128pp(gs[0]));
                    a[1] = gaussian_pair<DT>(xoshiro128pp(gs[0]));
                    *reinterpret_cast<u32x2 *>(abuf + (st & 1) * kABytes + ((rg * kSteps + ks) * 64 + lane) * 16 + 8 * hf) = a;
                }
            } else {
#pragma unroll
                for (int j = 0; j < kSteps / NH; ++j) {
                    const int ks = hf * (kSteps / NH) + j;
                    const u32x4 a = make_fragment(st, ks);
                    *reinterpret_cast<u32x4 *>(abuf + (st & 1) * kABytes + ((rg * kSteps + ks) * 64 + lane) * 16) = a;
                }
            }
        }
    };
    if constexpr (NH > 1) {
        publish_fragments(0, std::true_type{});
        __syncthreads();
    }
    // woven generator: the fragment of the step AFTER the current one is generated between the current step's MFMAs -- word + log of
    // pair q behind MFMA 2q, sqrt / cos / sin / pack behind MFMA 2q + 1 -- instead of as one clump of ~90 instructions in front of the
    // step, during which the wave feeds the matrix pipe nothing.  The SIMD's issue time is the same; the gain is the part of the clump
    // the wave's partner on the SIMD could not cover: 16384 x 768, p = 3276 bf16 116.8 -> 113.7 us, fp32 input 137.0 -> 130.1, 3072 wide
    // 428.6 -> 414.2 (profiles/r05_sketch_woven_ab.txt; scratch/gen_bench.hip had said -8 % for the bare loop).  The 4-wave tile
    // measured +-0.5 % and keeps the clump, as do the Rademacher kernels (8 instructions per fragment) and the two-half tile (its
    // waves generate for each other at different steps already).
    constexpr bool kWoven = kStreams && NH == 1 && W == 8 && NT == 8;
    u32x4 a_next = {0u, 0u, 0u, 0u};
    if constexpr (kWoven) {
        refresh_signs(0);
        a_next = make_fragment(0, 0);
    }
    // the multiply phase of one stage; FAST: the staging of stage s+1 (registers -> LDS) and the loads of stage s+2 are
    // unconditional and woven into the MFMA stream by the group barriers (one basic block)
    auto stage = [&](size_t s, auto fast_tag, auto first_tag) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_tag)::value;
        constexpr int first = decltype(first_tag)::value;      // the step whose slots carry the LDS writes (the loads follow one step later)
        uint8_t *cur = lds + (s & 1) * kStageBytes, *nxt = lds + ((s + 1) & 1) * kStageBytes;
        const uint8_t *next_base = stage_base + (s + 2) * stage_bytes;
        if constexpr (FAST) {
            finish_block<DT, false>(raw, blk, 8);      // (bf16 / fp16: nothing to do; fp32: the 16 v_cvt_pk_bf16_f32)
        } else {
            const int m1 = mode_of(s + 1), m2 = mode_of(s + 2);   // block-uniform
            stage_to_lds(s + 1, m1, nxt);
            fetch(s + 2, m2);
        }
        if constexpr (NH == 1 && !kWoven) refresh_signs(s);
        if constexpr (NH > 1 && !FAST) publish_fragments(s + 1, std::true_type{});
        __builtin_amdgcn_sched_barrier(0);
        // B fragments: all 8 of a 16-row step are in registers before its first MFMA, and each register is refilled with the
        // NEXT step's fragment right behind the MFMA that consumed it -- a full step (8 MFMAs = 256 cycles) of LDS latency cover.
        // hipcc's scheduler would sink every read to just in front of its MFMA (two registers, no cover) and cluster the staging
        // work in front of the MFMAs, so the order is pinned: one sched_barrier per MFMA slot, and inside a slot the order
        // written here.  FAST: slot t of step `first` carries the transpose + LDS write of feature t of stage s+1, slot t of
        // step `first + 1` the load of row t of stage s+2 (into the registers the transpose has just freed); with two column
        // halves the wave's share of the NEXT stage's A fragments is generated in front of step `first` as well -- the two waves
        // of a SIMD (same rows, other half) have different `first`, so one generates while the other multiplies.
        const uint8_t *frag = cur + (static_cast<size_t>(h) * BNT + 256 * hf + c) * 16;
        u32x4 bq[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bq[t] = *reinterpret_cast<const u32x4 *>(frag + 32 * 16 * t);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < kSteps; ++ks) {
            if constexpr (NH > 1 && FAST) {
                if (ks == first) {
                    publish_fragments(s + 1, std::false_type{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            u32x4 a;
            if constexpr (NH > 1) a = *reinterpret_cast<const u32x4 *>(abuf + (s & 1) * kABytes + ((rg * kSteps + ks) * 64 + lane) * 16);
            else if constexpr (kWoven) {
                a = a_next;                                    // generated between the MFMAs of the step before
                if (ks == kSteps - 1) {                        // the next step opens stage s + 1: its block's streams first, if it opens a block
                    refresh_signs(s + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            else a = make_fragment(s, ks);
            float wl2 = 0.0f, wu2 = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = Operand<DT>::mfma(a, bq[t], acc[t]);
                if (ks + 1 < kSteps)
                    bq[t] = *reinterpret_cast<const u32x4 *>(frag + (static_cast<size_t>(2 * (ks + 1)) * BNT + 32 * t) * 16);
                if constexpr (FAST) {
                    if (ks == first) store_feature<BNT>(blk, nxt, so, sfc, t);
                    if (ks == first + 1) raw.row[t] = load_raw<DT>(next_base + t * row_bytes + off0);
                }
                if constexpr (kWoven) {                        // dword t / 2 of the next step's fragment: word + log behind MFMA t, the rest behind t + 1
                    if ((t & 1) == 0) box_muller_begin(xoshiro128pp(gs[t >> 2]), wl2, wu2);
                    else a_next[t >> 1] = box_muller_end<DT>(wl2, wu2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    };
    // two loops, not one loop with a branch: the register allocator then sees the interior body (no branches, everything
    // pinned) on its own -- with both bodies in one loop it spilt half of the accumulators
    // (W = 8: the upper four waves do their staging four steps later than the lower four, so that the two waves of a SIMD are
    // not both in their VALU-heavy slots at once -- a wave-uniform choice between two copies of the loop, made once)
    typedef std::integral_constant<int, 0> Step0;
    typedef std::integral_constant<int, (W == 8 ? kSteps / 2 : 0)> StepMid;
    // (a static s_setprio 1 / 2 for the upper half measured 1 % slower, 161.2 / 161.4 against 159.7 us at 16384x3072, proj 1638)
    // (the same for the upper half of the 128 x 512 Gaussian tile: 3.5 % slower, 218.0 against 210.4 us; letting the two waves
    // of a SIMD take turns -- the younger one at priority 1 in every other step, or every other pair of steps -- changes nothing:
    // 159.6 / 159.1 against 160.5 us.  The older wave finishes a stage ~1600 cycles ahead and waits at the barrier, but the
    // SIMD's issue slots are busy either way)
    size_t s = 0;
    if constexpr (!RAGGED) {
        if (W == 8 && wave >= 4) {
            if constexpr (W == 8) for (; s + 2 < nfull; ++s) stage(s, std::true_type{}, StepMid{});
        } else {
            for (; s + 2 < nfull; ++s) stage(s, std::true_type{}, Step0{});
        }
    }
    for (; s < nstages; ++s) stage(s, std::false_type{}, Step0{});

    // ---- epilogue: accumulator register r of block t is S row (r&3) + 8*(r>>2) + 4*h, feature 8c + t
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const size_t i = m0 + 32 * rg + (r & 3) + 8 * (r >> 2) + 4 * h;
        const size_t f = n0 + 256 * hf + 8 * c;
        if (i >= proj || f >= features) continue;
        float v[8];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = acc[t][r];
        if constexpr (PARTIAL == 2) {                 // bf16 partial sums (bf16 results only: half the bytes of the slices' round trip)
            uint16_t *p = static_cast<uint16_t *>(out) + (static_cast<size_t>(bz) * proj + i) * features + f;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = Operand<FEWBIT_BF16>::pack(v[2 * e], v[2 * e + 1]);
            if (f + 8 <= features) {
                typedef u32x4 __attribute__((aligned(2))) u32x4u;
                *reinterpret_cast<u32x4u *>(p) = o;
            } else {
                for (int e = 0; e < 8; ++e) if (f + e < features) p[e] = static_cast<uint16_t>(o[e >> 1] >> (16 * (e & 1)));
            }
        } else if constexpr (PARTIAL == 1) {
            float *p = static_cast<float *>(out) + (static_cast<size_t>(bz) * proj + i) * features + f;
            if (f + 8 <= features) {
                typedef f32x4 __attribute__((aligned(4))) f32x4u;
                *reinterpret_cast<f32x4u *>(p) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4u *>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
                for (int e = 0; e < 8; ++e) if (f + e < features) p[e] = v[e];
            }
        } else if constexpr (DT == FEWBIT_F32) {
            float *p = static_cast<float *>(out) + i * features + f;
            if (f + 8 <= features) {
                typedef f32x4 __attribute__((aligned(4))) f32x4u;
                *reinterpret_cast<f32x4u *>(p) = f32x4{v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale};
                *reinterpret_cast<f32x4u *>(p + 4) = f32x4{v[4] * scale, v[5] * scale, v[6] * scale, v[7] * scale};
            } else {
                for (int e = 0; e < 8; ++e) if (f + e < features) p[e] = v[e] * scale;
            }
        } else {
            uint16_t *p = static_cast<uint16_t *>(out) + i * features + f;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = Operand<DT>::pack(v[2 * e] * scale, v[2 * e + 1] * scale);
            if (f + 8 <= features) {
                typedef u32x4 __attribute__((aligned(2))) u32x4u;
                *reinterpret_cast<u32x4u *>(p) = o;
            } else {
                for (int e = 0; e < 8; ++e) if (f + e < features) p[e] = static_cast<uint16_t>(o[e >> 1] >> (16 * (e & 1)));
            }
        }
    }
}

// partial sums -> result: fixed summation order (slice 0, 1, 2, ...), one scale, one rounding
template <int DT> __global__ __launch_bounds__(256) void sketch_reduce_kernel(const float *__restrict__ ws, size_t n, int slices, float scale, void *__restrict__ out) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    float s = ws[i];
    for (int z = 1; z < slices; ++z) s += ws[static_cast<size_t>(z) * n + i];
    s *= scale;
    if constexpr (DT == FEWBIT_F32) static_cast<float *>(out)[i] = s;
    else if constexpr (DT == FEWBIT_F16) static_cast<_Float16 *>(out)[i] = static_cast<_Float16>(s);
    else static_cast<__bf16 *>(out)[i] = static_cast<__bf16>(s);
}

// the same sums four elements per thread (n % 4 == 0): the loads of up to 8 slices are issued before the first add, so a thread
// has 8 x 16 bytes in flight instead of one dependent 4-byte load per slice (16384 x 768 bf16, p = 3276, 6 slices, rocprofv3: 15.2 -> 10.5 us)
template <int DT> __global__ __launch_bounds__(256) void sketch_reduce4_kernel(const float *__restrict__ ws, size_t n4, int slices, float scale, void *__restrict__ out) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(ws) + i;
    f32x4 s = src[0];
    int z = 1;
    for (; z + 8 <= slices; z += 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[static_cast<size_t>(z + k) * n4];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];                          // (slice order, as the scalar kernel)
    }
    if (z + 4 <= slices) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = src[static_cast<size_t>(z + k) * n4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s += v[k];
        z += 4;
    }
    {
        f32x4 v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = z + k < slices ? src[static_cast<size_t>(z + k) * n4] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 3; ++k) if (z + k < slices) s += v[k];      // (no `+ 0`: -0.0 stays -0.0, as in the scalar kernel)
    }
    s *= scale;
    if constexpr (DT == FEWBIT_F32) reinterpret_cast<f32x4 *>(out)[i] = s;
    else {
        typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
        u32x2 o;
        o.x = Operand<DT>::pack(s.x, s.y);
        o.y = Operand<DT>::pack(s.z, s.w);
        reinterpret_cast<u32x2 *>(out)[i] = o;
    }
}

// bf16 partial sums -> bf16 (or, OUT32, fp32) result: the same fixed order, sums in fp32, one scale, one rounding.  8 elements (16
// bytes) per thread and slice when n % 8 == 0 (VEC), else one element
template <bool VEC, bool OUT32> __global__ __launch_bounds__(256) void sketch_reduce_bf16_kernel(const uint16_t *__restrict__ ws, size_t n, int slices, float scale, void *__restrict__ out_) {
    typedef typename std::conditional<OUT32, float, uint16_t>::type out_t;
    out_t *out = static_cast<out_t *>(out_);
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    auto up = [](uint32_t hw) { return __builtin_bit_cast(float, hw << 16); };
    if constexpr (VEC) {
        const size_t n8 = n / 8;
        if (i >= n8) return;
        const u32x4 *src = reinterpret_cast<const u32x4 *>(ws) + i;
        float s[8];
        {
            const u32x4 v = src[0];
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[2 * e] = up(v[e] & 0xffffu); s[2 * e + 1] = up(v[e] >> 16); }
        }
        int z = 1;
        for (; z + 4 <= slices; z += 4) {
            u32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = src[static_cast<size_t>(z + k) * n8];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[2 * e] += up(v[k][e] & 0xffffu); s[2 * e + 1] += up(v[k][e] >> 16); }
        }
        for (; z < slices; ++z) {
            const u32x4 v = src[static_cast<size_t>(z) * n8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[2 * e] += up(v[e] & 0xffffu); s[2 * e + 1] += up(v[e] >> 16); }
        }
        if constexpr (OUT32) {
            f32x4 *o = reinterpret_cast<f32x4 *>(out) + 2 * i;
            o[0] = f32x4{s[0] * scale, s[1] * scale, s[2] * scale, s[3] * scale};
            o[1] = f32x4{s[4] * scale, s[5] * scale, s[6] * scale, s[7] * scale};
        } else {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = Operand<FEWBIT_BF16>::pack(s[2 * e] * scale, s[2 * e + 1] * scale);
            reinterpret_cast<u32x4 *>(out)[i] = o;
        }
    } else {
        if (i >= n) return;
        float s = up(ws[i]);
        for (int z = 1; z < slices; ++z) s += up(ws[static_cast<size_t>(z) * n + i]);
        if constexpr (OUT32) out[i] = s * scale;
        else out[i] = static_cast<uint16_t>(Operand<FEWBIT_BF16>::pack(s * scale, 0.0f) & 0xffffu);
    }
}

// the matrix itself (test seam and debugging aid; the product path never calls it): out[i][r] = S[row0 + i][col0 + r] as fp32
template <int DIST> __global__ __launch_bounds__(256) void sketch_matrix_kernel(Key key, size_t row0, size_t col0, size_t nrows, size_t ncols, int dtype, float *__restrict__ out) {
    const size_t idx = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= nrows * ncols) return;
    const size_t i = row0 + idx / ncols, r = col0 + idx % ncols;
    const int j = static_cast<int>(r & 7);
    float v;
    if constexpr (DIST == FEWBIT_SKETCH_RADEMACHER) {
        uint32_t w[4];
        const int s = static_cast<int>((r & 255) >> 4), h = static_cast<int>((r >> 3) & 1);
        philox4x32(static_cast<uint32_t>(i), static_cast<uint32_t>(2 * (r >> 8) + h), 0u, 0u, key, w);
        v = ((w[s >> 2] >> (((j & 1) ? 31 : 15) - (4 * (s & 3) + (j >> 1)))) & 1u) ? -1.0f : 1.0f;
    } else {
        float z0, z1;
        // element j of step s of block r / 256 for octet parity h: word 2 s + (q & 1) of stream q / 2, q = j / 2
        const int s = static_cast<int>((r & 255) >> 4), h = static_cast<int>((r >> 3) & 1), q = j >> 1;
        uint32_t st[4];
        philox4x32<kGaussianRounds>(static_cast<uint32_t>(i), static_cast<uint32_t>(2 * (r >> 8) + h), static_cast<uint32_t>(q >> 1), 2u, key, st);
        uint32_t w = 0;
        for (int k = 0; k <= 2 * s + (q & 1); ++k) w = xoshiro128pp(st);
        box_muller(w, z0, z1);
        v = (j & 1) ? z1 : z0;
        // rounded as the product kernel rounds its operand
        if (dtype == FEWBIT_F16) v = static_cast<float>(static_cast<_Float16>(v));
        else v = static_cast<float>(static_cast<__bf16>(v));
    }
    out[idx] = v;
}

// ---- fp32 input, many row tiles: one conversion pass first ------------------------------------------------------------------
// Every row tile of S re-reads M (from L2 / Infinity Cache).  For fp32 input that is twice the bytes of bf16 per re-read, twice
// the staging registers and a conversion per element per re-read; from a few row tiles on it is cheaper to round M to bf16 ONCE
// (a streaming pass: read 4 B, write 2 B per element, into the workspace) and run the bf16-input kernel on the copy, with the
// result still written as fp32 from the fp32 sums (measured, 16384 rows: p = 3276: -10 % at 768 features, -9...-19 % at 3072;
// p = 1638: -14 % / break-even; profiles/r04_sketch_preconvert.txt).  The products are the same numbers either way: both
// round M to bf16 with the same round-to-nearest-even conversion.
// piece i (8 consecutive features of one row) of the conversion
__device__ __forceinline__ void to_bf16_piece(size_t i, const float *__restrict__ m, size_t rows, size_t features, size_t ld, uint16_t *__restrict__ out) {
    const size_t per_row = (features + 7) / 8;
    if (i >= rows * per_row) return;
    const size_t r = i / per_row, f = (i % per_row) * 8;
    const float *src = m + r * ld + f;
    uint16_t *dst = out + r * features + f;
    if (f + 8 <= features) {
        typedef f32x4 __attribute__((aligned(4))) f32x4u;
        const f32x4 a = *reinterpret_cast<const f32x4u *>(src), b = *reinterpret_cast<const f32x4u *>(src + 4);
        u32x4 o;
        o[0] = Operand<FEWBIT_BF16>::pack(a[0], a[1]);
        o[1] = Operand<FEWBIT_BF16>::pack(a[2], a[3]);
        o[2] = Operand<FEWBIT_BF16>::pack(b[0], b[1]);
        o[3] = Operand<FEWBIT_BF16>::pack(b[2], b[3]);
        typedef u32x4 __attribute__((aligned(2))) u32x4u;
        *reinterpret_cast<u32x4u *>(dst) = o;
    } else {
        for (size_t e = 0; f + e < features; ++e) dst[e] = static_cast<uint16_t>(Operand<FEWBIT_BF16>::pack(src[e], 0.0f) & 0xffffu);
    }
}
__global__ __launch_bounds__(256) void to_bf16_kernel(const float *__restrict__ m, size_t rows, size_t features, size_t ld, uint16_t *__restrict__ out) {
    to_bf16_piece(static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x, m, rows, features, ld, out);
}

// ---- S written to memory once, in MFMA fragment order (the Gaussian sketch of layers wider than one column tile) --------------
// In the fused kernel every column tile of the grid regenerates the rows of S it multiplies: features / 256 times the generator's
// work, on the SIMD whose matrix pipe it shares -- measured (scratch/gen_bench.hip, profiles/r05_gen_bench.txt) a Gaussian fragment
// costs as many issue cycles as the 8 MFMAs it feeds, and VALU work beside an MFMA stream is not free beyond ~4 instructions per
// MFMA.  So for the Gaussian sketch of a layer wider than one tile, S is generated ONCE by this VALU-only kernel (every element
// once: ~11 us of VALU time for 3276 x 16384, in practice the ~20 us its 107 MB take to write) into the workspace, as the
// 16-byte-per-lane A fragments the product kernel's waves consume -- [32-row block of S][MFMA step][lane] -- and the product
// kernel (kFromMemory) reads them with one coalesced 1 KiB load per wave and step, kFragAhead steps ahead.  The same S as the
// fused kernel's (same streams, same Box-Muller, same rounding): which path ran is not visible in the result beyond the
// association of the slices' fp32 sums.  One wave = one 32-row block of S x one 256-row block of M = 16 fragments (16 KiB).
template <int DIST, int DT>
__device__ __forceinline__ void fragments_of_block(Key key, size_t b4, size_t rb, size_t nblocks, u32x4 *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const size_t b = b4 * 4 + wave;
    if (b >= nblocks) return;
    const uint32_t srow = static_cast<uint32_t>(32 * rb + c), blk = static_cast<uint32_t>(2 * b + h);
    u32x4 *dst = out + ((rb * nblocks + b) * 16) * 64 + lane;
    if constexpr (DIST == FEWBIT_SKETCH_RADEMACHER) {
        uint32_t signs[4];
        philox4x32(srow, blk, 0u, 0u, key, signs);
#pragma unroll
        for (int st = 0; st < 16; ++st) dst[st * 64] = rademacher_fragment<DT>(signs, st);
    } else {
        uint32_t g0[4], g1[4];
        philox4x32<kGaussianRounds>(srow, blk, 0u, 2u, key, g0);
        philox4x32<kGaussianRounds>(srow, blk, 1u, 2u, key, g1);
#pragma unroll 4
        for (int st = 0; st < 16; ++st) {
            u32x4 a;
            a[0] = gaussian_pair<DT>(xoshiro128pp(g0));
            a[1] = gaussian_pair<DT>(xoshiro128pp(g0));
            a[2] = gaussian_pair<DT>(xoshiro128pp(g1));
            a[3] = gaussian_pair<DT>(xoshiro128pp(g1));
            dst[st * 64] = a;
        }
    }
}
template <int DIST, int DT>
__global__ __launch_bounds__(256) void sketch_fragments_kernel(Key key, const Key *__restrict__ key_dev, size_t nblocks, u32x4 *__restrict__ out) {
    if (key_dev != nullptr) key = *key_dev;
    fragments_of_block<DIST, DT>(key, blockIdx.x, blockIdx.y, nblocks, out);
}
// ---- seeds drawn on the device ----------------------------------------------------------------------------------------------
// A launch recorded in a hipGraph replays its kernel ARGUMENTS: a seed passed by value would give every replay the same S.
// There the seed comes from device memory instead: `next_seed_kernel` (also recorded) bumps a counter and derives the seed of
// this call from it, so every replay -- and every call inside one replay -- draws a fresh matrix; the backward of the layer
// reads the seed its forward left behind.  mix_seed = splitmix64's finaliser over base + (count + 1) * golden ratio.
__host__ __device__ __forceinline__ uint64_t mix_seed(uint64_t base, uint64_t count) {
    uint64_t x = base + (count + 1) * 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void next_seed_kernel(uint64_t *counter, uint64_t base, uint64_t *seed) {
    // (an atomic bump: one counter per device serves every captured graph, and two graphs may replay on two streams at once)
    const uint64_t count = atomicAdd(reinterpret_cast<unsigned long long *>(counter), 1ull);
    *seed = mix_seed(base, count);
}

// ---- host side --------------------------------------------------------------------------------------------------------------
struct Plan { unsigned gx, gy, gz; size_t kslice; int waves, halves; };
struct Seed { Key value; const Key *device; };       // `device` != nullptr: the key is read from there when the kernel runs
struct Frags { const void *data; size_t steps; };    // A fragments in memory (kFromMemory): `steps` MFMA steps per 32-row block of S

int device_cus() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

FEWBIT_HIDDEN std::atomic<long long> g_forced_slices{-1}, g_forced_waves{-1}, g_forced_halves{-1};

// K slices: the tile grid of a sketch is small (proj x features) and K = rows is long, so the rows are cut into `gz` slices
// when that fills the chip better.  cost(z) = (rounds of the CU array with z x tiles workgroups) / z, plus 4 % per extra
// slice for the partial-sum traffic; slices are multiples of 256 rows (the Rademacher block) and at least 1024 rows.
// `slots` = workgroups the chip holds at once: 2 per CU for the 4-wave tile, 1 per CU for the 8-wave tile.
double plan_slices(size_t tiles, size_t slots, size_t rows, long long forced, size_t &best) {
    size_t max_z = rows / 1024;
    if (max_z < 1) max_z = 1;
    if (max_z > 16) max_z = 16;
    best = 1;
    double best_cost = 1e30;
    for (size_t z = 1; z <= max_z; ++z) {
        if (forced > 0 && z != static_cast<size_t>(forced) && !(static_cast<size_t>(forced) > max_z && z == max_z)) continue;
        const size_t units = tiles * z;
        const double rounds = static_cast<double>((units + slots - 1) / slots);
        const double cost = rounds / static_cast<double>(z) * (1.0 + 0.04 * static_cast<double>(z - 1));
        if (cost < best_cost - 1e-12) { best_cost = cost; best = z; }
    }
    return best_cost;
}

Plan make_plan(int dist, int dtype, size_t rows, size_t features, size_t proj, bool one_half = false) {
    const long long forced_z = g_forced_slices.load(std::memory_order_relaxed), forced_w = g_forced_waves.load(std::memory_order_relaxed);
    const long long forced_h = g_forced_halves.load(std::memory_order_relaxed);
    const size_t cus = static_cast<size_t>(device_cus());
    Plan p;
    // the Gaussian sketch is bound by the generator (instruction issue): the 128 x 512 tile generates every element of S once per
    // 512 columns instead of once per 256 (two column halves share their A fragments through LDS).  Measured (16384 rows,
    // bf16): 3072 features +11-16 % at p = 1638 / 3276 / 8192; 768 features (one and a half wide tiles) -8 %; fp32 input no gain
    // (its staging registers already spill) -- so: 16-bit input, and a feature count the 512-wide tile divides or >= 2048
    bool wide = dist == FEWBIT_SKETCH_GAUSSIAN && dtype != FEWBIT_F32 && features >= 1024 && (features % (2 * BN) == 0 || features >= 2048);
    if (forced_h == 1) wide = false;
    if (forced_h == 2) wide = true;
    if (one_half) wide = false;                      // (fragments from memory: the product kernel has the one-half tiles only)
    if (wide) {
        p.waves = 8;
        p.halves = 2;
        p.gx = static_cast<unsigned>((features + 2 * BN - 1) / (2 * BN));
        p.gy = static_cast<unsigned>((proj + 127) / 128);
        size_t z = 1;
        plan_slices(static_cast<size_t>(p.gx) * p.gy, cus, rows, forced_z, z);
        size_t kslice = (rows + z - 1) / z;
        kslice = (kslice + 255) / 256 * 256;
        p.kslice = kslice;
        p.gz = static_cast<unsigned>((rows + kslice - 1) / kslice);
        if (p.gz < 1) p.gz = 1;
        return p;
    }
    p.halves = 1;
    p.gx = static_cast<unsigned>((features + BN - 1) / BN);
    // tile height: 256 rows of S (8 waves) when that does not waste more of the last row tile than 128 rows (4 waves) would
    // cost in staging
    size_t z4 = 1, z8 = 1;
    const size_t t4 = static_cast<size_t>(p.gx) * ((proj + 127) / 128), t8 = static_cast<size_t>(p.gx) * ((proj + 255) / 256);
    // (a round of 2 x CUs short tiles and a round of CUs tall tiles are the same MFMA work per CU; measured, the tall tile runs
    // 3-10 % faster per MFMA: half the staging work and L2 reads)
    const double c4 = plan_slices(t4, 2 * cus, rows, forced_z, z4);
    const double c8 = plan_slices(t8, cus, rows, forced_z, z8) * 0.93;
    bool tall = c8 <= c4;
    if (forced_w == 4) tall = false;
    if (forced_w == 8) tall = true;
    p.waves = tall ? 8 : 4;
    p.gy = static_cast<unsigned>((proj + (tall ? 255 : 127)) / (tall ? 256 : 128));
    const size_t z = tall ? z8 : z4;
    size_t kslice = (rows + z - 1) / z;
    kslice = (kslice + 255) / 256 * 256;
    p.kslice = kslice;
    p.gz = static_cast<unsigned>((rows + kslice - 1) / kslice);
    if (p.gz < 1) p.gz = 1;
    return p;
}

template <int DIST, int DT, int PARTIAL, int W, int NH = 1>
int launch_kernel(const Plan &p, bool ragged, const void *m, size_t rows, size_t features, size_t ld, size_t proj, Seed key, float scale, void *out,
                  Frags frags, hipStream_t s) {
    const dim3 grid(p.gx, p.gy, p.gz), block(Tile<W, NH>::kThreads);
    constexpr size_t lds = Tile<W, NH>::kLdsBytes;
    auto go = [&](auto kern) -> int {
        if (lds > 65536) {                           // (more than the default limit of a workgroup: opt in once per kernel and device)
            // (one flag word per KERNEL: the ragged and the plain variant have the same function type, hence share this lambda's
            // instantiation -- they are told apart by index)
            static std::atomic<unsigned long long> done_flags[2];
            std::atomic<unsigned long long> &done = done_flags[ragged ? 1 : 0];
            int dev = 0;
            (void)hipGetDevice(&dev);
            const unsigned long long bit = 1ull << (dev & 63);
            if (!(done.load(std::memory_order_relaxed) & bit)) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
                    (void)hipGetLastError();
                    return fail(FEWBIT_ERR_LAUNCH, "sketch: cannot reserve %zu bytes of LDS", lds);
                }
                done.fetch_or(bit, std::memory_order_relaxed);
            }
        }
        hipLaunchKernelGGL(kern, grid, block, lds, s, m, rows, features, ld, proj, key.value, key.device, scale, out, p.kslice, frags.data, frags.steps);
        return FEWBIT_OK;
    };
    return ragged ? go(sketch_kernel<DIST, DT, PARTIAL, true, W, NH>) : go(sketch_kernel<DIST, DT, PARTIAL, false, W, NH>);
}

// bf16 partial sums: when the rows are sliced and the operands are bf16, the slices' round trip through memory (written by
// the product kernel, read back by the reduce kernel -- 2 x gz x proj x features x 4 bytes, the K-independent part of a 768-wide
// product's time) is made in bf16: each slice's sum is rounded once, the slices are added in fp32 in the same fixed order.
//   * bf16 result: its error grows from one bf16 rounding to about sqrt(2) of one (gz roundings of sums sqrt(gz) times smaller);
//   * fp32 result of an fp32 input that was rounded to bf16 first: the sums already carry the operands' rounding (2^-9 per
//     element of M -- and of a Gaussian S --: ~0.6-0.8 x 2^-9 of a slice's sum in its standard deviation); rounding the slice's sum to
//     bf16 adds ~0.4 x 2^-9 of it, +12 % on that error, under an estimator whose own relative noise is sqrt(rows / p).
// fp16 keeps fp32 partial sums (range).  tune: 0 never, 2 for bf16 results only, 1 / -1 this policy.
FEWBIT_HIDDEN std::atomic<long long> g_forced_partial16{-1};
bool partial16(int dtype, int out_dtype, unsigned gz) {
    if (dtype != FEWBIT_BF16 || gz <= 1 || (out_dtype != FEWBIT_BF16 && out_dtype != FEWBIT_F32)) return false;
    const long long forced = g_forced_partial16.load(std::memory_order_relaxed);
    return forced != 0 && !(forced == 2 && out_dtype != FEWBIT_BF16);
}

template <int DIST, int DT, int PARTIAL>
int launch_tile(const Plan &p, bool ragged, const void *m, size_t rows, size_t features, size_t ld, size_t proj, Seed key, float scale, void *out,
                Frags frags, hipStream_t s) {
    if constexpr (DIST != kFromMemory) {
        if (p.halves == 2) return launch_kernel<DIST, DT, PARTIAL, 8, 2>(p, ragged, m, rows, features, ld, proj, key, scale, out, frags, s);
    }
    return p.waves == 8 ? launch_kernel<DIST, DT, PARTIAL, 8>(p, ragged, m, rows, features, ld, proj, key, scale, out, frags, s)
                        : launch_kernel<DIST, DT, PARTIAL, 4>(p, ragged, m, rows, features, ld, proj, key, scale, out, frags, s);
}

template <int DIST, int DT>
int launch(const void *m, size_t rows, size_t features, size_t ld, size_t proj, Seed key, float scale, void *out, int out_dtype, void *workspace,
           size_t workspace_bytes, Frags frags, hipStream_t s) {
    const Plan p = make_plan(DIST == kFromMemory ? static_cast<int>(FEWBIT_SKETCH_RADEMACHER) : DIST, DT, rows, features, proj, DIST == kFromMemory);
    const bool ragged = (features % 8) != 0;
    int rc;
    if (p.gz == 1 && out_dtype == DT) {
        rc = launch_tile<DIST, DT, 0>(p, ragged, m, rows, features, ld, proj, key, scale, out, frags, s);
        if (rc != FEWBIT_OK) return rc;
    } else {                                         // partial sums, then one pass: sum, scale, round to the result's dtype
        const bool p16 = partial16(DT, out_dtype, p.gz);
        const size_t n = proj * features;
        const size_t need = static_cast<size_t>(p.gz) * n * (p16 ? sizeof(uint16_t) : sizeof(float));
        if (workspace == nullptr || workspace_bytes < need)
            return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch: workspace of %zu bytes needed (fewbit_hip_sketch_workspace), got %zu", need, workspace_bytes);
        const int z = static_cast<int>(p.gz);
        if constexpr (DT == FEWBIT_BF16) {
            if (p16) {
                rc = launch_tile<DIST, DT, 2>(p, ragged, m, rows, features, ld, proj, key, scale, workspace, frags, s);
                if (rc != FEWBIT_OK) return rc;
                const uint16_t *ws = static_cast<const uint16_t *>(workspace);
                const bool vec = n % 8 == 0 && (reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(out)) % 16 == 0;
                const dim3 grid(static_cast<unsigned>(((vec ? n / 8 : n) + 255) / 256));
                if (out_dtype == FEWBIT_F32) {
                    if (vec) hipLaunchKernelGGL((sketch_reduce_bf16_kernel<true, true>), grid, dim3(256), 0, s, ws, n, z, scale, out);
                    else hipLaunchKernelGGL((sketch_reduce_bf16_kernel<false, true>), grid, dim3(256), 0, s, ws, n, z, scale, out);
                } else {
                    if (vec) hipLaunchKernelGGL((sketch_reduce_bf16_kernel<true, false>), grid, dim3(256), 0, s, ws, n, z, scale, out);
                    else hipLaunchKernelGGL((sketch_reduce_bf16_kernel<false, false>), grid, dim3(256), 0, s, ws, n, z, scale, out);
                }
                const hipError_t e = hipGetLastError();
                if (e != hipSuccess) return fail(FEWBIT_ERR_LAUNCH, "sketch: %s", hipGetErrorString(e));
                return FEWBIT_OK;
            }
        }
        rc = launch_tile<DIST, DT, 1>(p, ragged, m, rows, features, ld, proj, key, scale, workspace, frags, s);
        if (rc != FEWBIT_OK) return rc;
        const float *ws = static_cast<const float *>(workspace);
        const bool vec = n % 4 == 0 && (reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(out)) % 16 == 0;
        auto reduce = [&](auto tag) {
            constexpr int ODT = decltype(tag)::value;
            if (vec) hipLaunchKernelGGL((sketch_reduce4_kernel<ODT>), dim3(static_cast<unsigned>((n / 4 + 255) / 256)), dim3(256), 0, s, ws, n / 4, z, scale, out);
            else hipLaunchKernelGGL((sketch_reduce_kernel<ODT>), dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, ws, n, z, scale, out);
        };
        if (out_dtype == FEWBIT_F32) reduce(std::integral_constant<int, FEWBIT_F32>{});
        else reduce(std::integral_constant<int, DT>{});
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FEWBIT_ERR_LAUNCH, "sketch: %s", hipGetErrorString(e));
    return FEWBIT_OK;
}

template <int DIST>
int launch_dtype(int dtype, const void *m, size_t rows, size_t features, size_t ld, size_t proj, Seed key, float scale, void *out, int out_dtype,
                 void *workspace, size_t workspace_bytes, Frags frags, hipStream_t s) {
    switch (dtype) {
    case FEWBIT_F32:
        if constexpr (DIST == kFromMemory) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch: fragments from memory need a 16-bit operand");
        else return launch<DIST, FEWBIT_F32>(m, rows, features, ld, proj, key, scale, out, out_dtype, workspace, workspace_bytes, frags, s);
    case FEWBIT_F16: return launch<DIST, FEWBIT_F16>(m, rows, features, ld, proj, key, scale, out, out_dtype, workspace, workspace_bytes, frags, s);
    case FEWBIT_BF16: return launch<DIST, FEWBIT_BF16>(m, rows, features, ld, proj, key, scale, out, out_dtype, workspace, workspace_bytes, frags, s);
    default: return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch: unknown dtype %d", dtype);
    }
}

// fp32 input: convert first?  From 6 row tiles of 256 on (p > 1280) -- below that M is read too few times for the extra pass
// (read 4 + write 2 + read 2 bytes per element) to pay.  tune: 0 never, 1 always, -1 this policy.
FEWBIT_HIDDEN std::atomic<long long> g_forced_convert{-1};
bool converts_first(int dtype, size_t rows, size_t proj) {
    if (dtype != FEWBIT_F32 || rows == 0) return false;
    const long long forced = g_forced_convert.load(std::memory_order_relaxed);
    if (forced >= 0) return forced != 0;
    return proj > 1280;
}
constexpr size_t kWorkspaceAlign = 256;
size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// S in memory first?  The Gaussian sketch (a fragment costs ~100 issue slots; Rademacher's 8 are not worth a byte of traffic) when
// at least two column tiles would otherwise regenerate it (features > 256) and its fragments (bf16, rows of S padded to 256, rows
// of M to 256) stay under 1 GiB -- for a 16-bit INPUT always, for fp32 input that is rounded to bf16 first only on layers at least
// kWideLayer features wide: stand-alone it is 5-15 % faster on narrow fp32 layers too, but inside an fp32 model every from-memory
// product slowed the REST of the step (full-entropy operands at the matrix pipe's highest duty pull the shader clock down) by more
// than a 768-wide product gains and less than a 3072-wide one gains (profiles/r05_roberta_ab_width.txt; EXPERIMENTS.md).
// tune "sketch_materialise": 0 never, -1 this policy, 1 the policy without the width rule of fp32 input (the caps stay).
constexpr size_t kWideLayer = 2048;
constexpr size_t kMaxFragmentBytes = 1ull << 30;
FEWBIT_HIDDEN std::atomic<long long> g_forced_materialise{-1};
size_t fragment_blocks(size_t rows) { return (rows + 255) / 256; }                  // 256-row blocks of M = 16 MFMA steps each
size_t fragment_row_blocks(size_t proj) { return (proj + 255) / 256 * 8; }          // 32-row blocks of S, padded to whole 256-row tiles
size_t fragment_bytes(size_t rows, size_t proj) { return (fragment_row_blocks(proj) * fragment_blocks(rows) * 16 + kFragAhead) * 1024; }
bool materialises(int dist, int operand_dtype, bool converted, size_t rows, size_t features, size_t proj) {
    if (dist != FEWBIT_SKETCH_GAUSSIAN || operand_dtype == FEWBIT_F32 || rows == 0) return false;
    const long long forced = g_forced_materialise.load(std::memory_order_relaxed);
    if (forced == 0 || fragment_row_blocks(proj) > 65535) return false;
    if (features <= BN || fragment_bytes(rows, proj) > kMaxFragmentBytes) return false;      // (also when forced: the path exists to SAVE time and memory)
    return forced == 1 || !converted || features >= kWideLayer;
}

// the workspace of one call: [partial sums][bf16 copy of an fp32 M][A fragments of S], each part aligned to kWorkspaceAlign
struct Layout { int operand_dtype, plan_dist; bool converted, materialised; size_t partial_bytes, copy_off, copy_bytes, frag_off, frag_bytes, total; };
Layout layout(int dist, int dtype, size_t rows, size_t features, size_t proj) {
    Layout L{};
    L.converted = converts_first(dtype, rows, proj);
    L.operand_dtype = L.converted ? static_cast<int>(FEWBIT_BF16) : dtype;
    L.materialised = materialises(dist, L.operand_dtype, L.converted, rows, features, proj);
    L.plan_dist = L.materialised ? static_cast<int>(FEWBIT_SKETCH_RADEMACHER) : dist;      // (fragments from memory: the one-half tiles' plan)
    const Plan p = make_plan(L.plan_dist, L.operand_dtype, rows, features, proj, L.materialised);
    L.partial_bytes = (p.gz > 1 || dtype != L.operand_dtype)
                          ? static_cast<size_t>(p.gz) * proj * features * (partial16(L.operand_dtype, dtype, p.gz) ? sizeof(uint16_t) : sizeof(float)) : 0;
    size_t end = L.partial_bytes;
    if (L.converted) { L.copy_off = round_up(end, kWorkspaceAlign); L.copy_bytes = rows * features * sizeof(uint16_t); end = L.copy_off + L.copy_bytes; }
    if (L.materialised) { L.frag_off = round_up(end, kWorkspaceAlign); L.frag_bytes = fragment_bytes(rows, proj); end = L.frag_off + L.frag_bytes; }
    L.total = end;
    return L;
}

}  // namespace sketch
}  // namespace fewbit_hip

namespace fewbit_hip {
// the keys of fewbit_hip_tune (fewbit_kernels.hip) that belong to this unit; *known = false for any other key
FEWBIT_HIDDEN int sketch_tune(const char *key, long long value, bool *known) {
    using namespace sketch;
    struct Spec { const char *key; std::atomic<long long> *slot; bool (*valid)(long long); const char *what; };
    static const Spec specs[] = {
        {"sketch_slices", &g_forced_slices, [](long long v) { return v == -1 || v > 0; }, "the number of row slices (> 0), -1 = policy"},
        {"sketch_waves", &g_forced_waves, [](long long v) { return v == -1 || v == 4 || v == 8; }, "waves per workgroup: 4 (128-row tile), 8 (256-row tile), -1 = policy"},
        {"sketch_halves", &g_forced_halves, [](long long v) { return v == -1 || v == 1 || v == 2; }, "column halves per workgroup: 1, 2 (the 128 x 512 tile), -1 = policy"},
        {"sketch_convert", &g_forced_convert, [](long long v) { return v >= -1 && v <= 1; }, "fp32 input rounded to bf16 in one pass first: 0 never, 1 always, -1 = policy"},
        {"sketch_partials", &g_forced_partial16, [](long long v) { return v >= -1 && v <= 2; }, "bf16 partial sums: 0 never, 2 for bf16 results only, 1 / -1 = policy"},
        {"sketch_materialise", &g_forced_materialise, [](long long v) { return v >= -1 && v <= 1; }, "Gaussian S through memory: 0 never, 1 also on narrow fp32 layers, -1 = policy"},
    };
    *known = false;
    for (const Spec &sp : specs) {
        if (strcmp(key, sp.key) != 0) continue;
        *known = true;
        if (!sp.valid(value)) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "tune: %s = %lld; expected %s", key, value, sp.what);
        sp.slot->store(value, std::memory_order_relaxed);
        return FEWBIT_OK;
    }
    return FEWBIT_OK;
}
}  // namespace fewbit_hip

using namespace fewbit_hip;
using namespace fewbit_hip::sketch;

extern "C" {

size_t fewbit_hip_sketch_workspace(int dist, int dtype, size_t rows, size_t features, size_t proj) {
    if (rows == 0 || features == 0 || proj == 0) return 0;
    return layout(dist, dtype, rows, features, proj).total;
}

static int sketch_entry(int dist, int dtype, const void *m, size_t rows, size_t features, size_t ld, size_t proj, Seed key, double scale,
                        void *out, void *workspace, size_t workspace_bytes, void *stream) {
    if (dist != FEWBIT_SKETCH_RADEMACHER && dist != FEWBIT_SKETCH_GAUSSIAN) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch: unknown distribution %d", dist);
    if (proj == 0 || features == 0) return FEWBIT_OK;
    if (out == nullptr || (m == nullptr && rows != 0)) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch: null pointer");
    if (ld < features) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch: leading dimension %zu < features %zu", ld, features);
    if (proj > 0xffffffffull || (rows >> 3) > 0xffffffffull) return fail(FEWBIT_ERR_UNSUPPORTED, "sketch: proj and rows/8 must fit 32 bits");
    if ((ld + 1) * 72 * 4 >= 0x80000000ull) return fail(FEWBIT_ERR_UNSUPPORTED, "sketch: leading dimension %zu too large (a K stage must span less than 2 GiB)", ld);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rows == 0) {                                  // empty sum: zeros
        const size_t es = dtype == FEWBIT_F32 ? 4 : 2;
        if (hipMemsetAsync(out, 0, proj * features * es, s) != hipSuccess) return fail(FEWBIT_ERR_LAUNCH, "sketch: memset failed");
        return FEWBIT_OK;
    }
    const Layout L = layout(dist, dtype, rows, features, proj);
    if (L.total != 0 && (workspace == nullptr || workspace_bytes < L.total))
        return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch: workspace of %zu bytes needed (fewbit_hip_sketch_workspace), got %zu", L.total, workspace_bytes);
    uint8_t *ws = static_cast<uint8_t *>(workspace);
    auto launch_fragments = [&]() {
        const size_t nblocks = fragment_blocks(rows);
        const dim3 grid(static_cast<unsigned>((nblocks + 3) / 4), static_cast<unsigned>(fragment_row_blocks(proj)));
        u32x4 *frag = reinterpret_cast<u32x4 *>(ws + L.frag_off);
        if (L.operand_dtype == FEWBIT_F16) hipLaunchKernelGGL((sketch_fragments_kernel<FEWBIT_SKETCH_GAUSSIAN, FEWBIT_F16>), grid, dim3(256), 0, s, key.value, key.device, nblocks, frag);
        else hipLaunchKernelGGL((sketch_fragments_kernel<FEWBIT_SKETCH_GAUSSIAN, FEWBIT_BF16>), grid, dim3(256), 0, s, key.value, key.device, nblocks, frag);
    };
    if (L.converted) {                                // fp32 input, many row tiles: rounded to bf16 once
        uint16_t *copy = reinterpret_cast<uint16_t *>(ws + L.copy_off);
        const size_t pieces = rows * ((features + 7) / 8);
        hipLaunchKernelGGL(to_bf16_kernel, dim3(static_cast<unsigned>((pieces + 255) / 256)), dim3(256), 0, s, static_cast<const float *>(m), rows, features, ld, copy);
        m = copy;
        ld = features;
    }
    const float fscale = static_cast<float>(scale);
    if (L.materialised) {                             // S once, as A fragments; then the product kernel that reads them
        const size_t nblocks = fragment_blocks(rows);
        u32x4 *frag = reinterpret_cast<u32x4 *>(ws + L.frag_off);
        launch_fragments();       // (behind the conversion pass of an fp32 input: in front of it measured the same)
        // (the kFragAhead steps of padding behind the last fragment are read, never used: any bytes will do)
        return launch_dtype<kFromMemory>(L.operand_dtype, m, rows, features, ld, proj, key, fscale, out, dtype, workspace, L.partial_bytes, Frags{frag, nblocks * 16}, s);
    }
    if (dist == FEWBIT_SKETCH_RADEMACHER)
        return launch_dtype<FEWBIT_SKETCH_RADEMACHER>(L.operand_dtype, m, rows, features, ld, proj, key, fscale, out, dtype, workspace, L.partial_bytes, Frags{nullptr, 0}, s);
    return launch_dtype<FEWBIT_SKETCH_GAUSSIAN>(L.operand_dtype, m, rows, features, ld, proj, key, fscale, out, dtype, workspace, L.partial_bytes, Frags{nullptr, 0}, s);
}

int fewbit_hip_sketch(int dist, int dtype, const void *m, size_t rows, size_t features, size_t ld, size_t proj, uint64_t seed, double scale,
                      void *out, void *workspace, size_t workspace_bytes, void *stream) {
    const Seed key{Key{static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)}, nullptr};
    return sketch_entry(dist, dtype, m, rows, features, ld, proj, key, scale, out, workspace, workspace_bytes, stream);
}

int fewbit_hip_sketch_device_seed(int dist, int dtype, const void *m, size_t rows, size_t features, size_t ld, size_t proj,
                                  const uint64_t *seed_device, double scale, void *out, void *workspace, size_t workspace_bytes, void *stream) {
    if (seed_device == nullptr) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch: null seed pointer");
    static_assert(sizeof(Key) == sizeof(uint64_t), "a Key is the two halves of the 64-bit seed, low word first");
    const Seed key{Key{0u, 0u}, reinterpret_cast<const Key *>(seed_device)};
    return sketch_entry(dist, dtype, m, rows, features, ld, proj, key, scale, out, workspace, workspace_bytes, stream);
}

uint64_t fewbit_hip_sketch_mix_seed(uint64_t base, uint64_t count) { return mix_seed(base, count); }

int fewbit_hip_sketch_next_seed(uint64_t *counter_device, uint64_t base, uint64_t *seed_device, void *stream) {
    if (counter_device == nullptr || seed_device == nullptr) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch_next_seed: null pointer");
    hipLaunchKernelGGL(next_seed_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), counter_device, base, seed_device);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FEWBIT_ERR_LAUNCH, "sketch_next_seed: %s", hipGetErrorString(e));
    return FEWBIT_OK;
}

int fewbit_hip_sketch_matrix(int dist, int dtype, uint64_t seed, size_t row0, size_t col0, size_t nrows, size_t ncols, float *out, void *stream) {
    if (dist != FEWBIT_SKETCH_RADEMACHER && dist != FEWBIT_SKETCH_GAUSSIAN) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch: unknown distribution %d", dist);
    if (nrows == 0 || ncols == 0) return FEWBIT_OK;
    if (out == nullptr) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch: null pointer");
    const Key key{static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)};
    const size_t n = nrows * ncols;
    const dim3 grid(static_cast<unsigned>((n + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dist == FEWBIT_SKETCH_RADEMACHER) hipLaunchKernelGGL((sketch_matrix_kernel<FEWBIT_SKETCH_RADEMACHER>), grid, dim3(256), 0, s, key, row0, col0, nrows, ncols, dtype, out);
    else hipLaunchKernelGGL((sketch_matrix_kernel<FEWBIT_SKETCH_GAUSSIAN>), grid, dim3(256), 0, s, key, row0, col0, nrows, ncols, dtype, out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FEWBIT_ERR_LAUNCH, "sketch_matrix: %s", hipGetErrorString(e));
    return FEWBIT_OK;
}

int fewbit_hip_sketch_describe(int dist, int dtype, size_t rows, size_t features, size_t proj, char *buf, size_t len) {
    if (buf == nullptr || len == 0) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "sketch_describe: no buffer");
    const Layout L = layout(dist, dtype, rows, features, proj);
    const Plan p = make_plan(L.plan_dist, L.operand_dtype, rows, features, proj, L.materialised);
    const char *partials = (p.gz > 1 || L.operand_dtype != dtype) ? (partial16(L.operand_dtype, dtype, p.gz) ? "\"bf16\"" : "\"fp32\"") : "null";
    snprintf(buf, len, "{\"kernel\": \"sketch_kernel (%dx%d tile, K stage %d, v_mfma_f32_32x32x16%s)\", \"grid\": [%u, %u, %u], \"threads\": %d, "
                       "\"k_slice\": %zu, \"lds_bytes\": %d, \"workspace_bytes\": %zu, \"converted_to_bf16_first\": %s, \"partial_sums\": %s, "
                       "\"s_fragment_bytes\": %zu}",
             32 * p.waves / p.halves, 256 * p.halves, 16 * p.waves / p.halves, L.materialised ? ", A fragments of S from memory" : "", p.gx, p.gy, p.gz,
             64 * p.waves, p.kslice, p.halves == 2 ? Tile<8, 2>::kLdsBytes : 2 * 16 * p.waves * BN * 2, rows == 0 ? 0 : L.total,
             L.converted ? "true" : "false", partials, L.frag_bytes);
    return FEWBIT_OK;
}

void fewbit_hip_philox4x32(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t o[4];
    philox4x32(counter[0], counter[1], counter[2], counter[3], Key{key[0], key[1]}, o);
    for (int i = 0; i < 4; ++i) out[i] = o[i];
}

void fewbit_hip_xoshiro128pp(uint32_t state[4], uint32_t *out, size_t n) {
    uint32_t s[4] = {state[0], state[1], state[2], state[3]};
    for (size_t i = 0; i < n; ++i) out[i] = xoshiro128pp(s);
    for (int i = 0; i < 4; ++i) state[i] = s[i];
}

}  // extern "C"
