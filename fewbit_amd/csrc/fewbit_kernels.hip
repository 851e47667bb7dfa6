// fewbit_kernels.hip -- gfx950 kernels + the C-ABI (include/fewbit_hip.h) of FewBit's
// quantized-activation path.  From-scratch CDNA4 design; what each kernel replaces in the reference
// (skolai/fewbit) is cited at its definition.
//
// Work decomposition:
//   group      = 8 consecutive elements  <-> exactly K bytes of packed state
//   wave tile  = 64 lanes x U groups; load u of a wave covers 64 lane-contiguous groups, so every
//                wave-level memory instruction on x / y / gy / gx touches one contiguous span
//                (1 KiB for 16-bit dtypes)
//   grid       = ONE resident generation of waves (occupancy API x #CUs); wave w loops over tiles
//                w, w + W, w + 2W, ... with a two-buffer software pipeline (pipeline2): the loads of
//                the next tile are in flight while the current one is processed
// Kernels:
//   quantize_forward_lut_kernel   fp16/bf16, K <= 4, large tensors: code = byte table in LDS indexed by
//                                 the raw 16-bit pattern (built per 1024-thread block)
//   quantize_forward_kernel       fp32, and 16-bit below the table threshold: borders in wave-uniform
//                                 VGPRs (v_readlane), MSB-first search in hand-scheduled asm
//   quantize_backward_kernel      levels staged in LDS, one gather + one multiply per element
//   stepwise1_*_kernel            the exact 1-bit family
//   quantize_*_wide_kernel        tables of 5..8 bits and tables that do not fill their bit width
// No kernel needs aligned pointers: gfx950 global accesses take any address (see fewbit_device.h).
//   pack/unpack_codes_kernel      codec seam used by the tests
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "fewbit_codepack.h"
#include "fewbit_device.h"

namespace fewbit_hip {

// ------------------------------------------------------------------------------------------------
// code of one element against NB sorted borders (edge path only; the fast path uses pack_group):
//   code = #{ j : b_j < x }, and NaN -> NB because the predicate is !(b_j >= x)
// which is torch.searchsorted's CPU rule used by the reference (fewbit/cpu/gelu.cc:17).
template <int NB> __device__ __forceinline__ uint32_t count_below(const float (&b)[NB], float x) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) c += !(b[j] >= x) ? 1u : 0u;
    return c;
}

// lanes 0..NB-1 fetch one border each (fetch_border: issued before the first tile so that waiting for it -- vmcnt
// counts in order -- does not wait for the tile too); every lane then holds all NB (wave-uniform) values
template <int DT, int NB> __device__ __forceinline__ float fetch_border(const void *borders) {
    const int lane = threadIdx.x & (kWave - 1);
    float mine = 0.0f;
    if (lane < NB) mine = Elem<DT>::load(borders, lane);
    return mine;
}
template <int NB> __device__ __forceinline__ void spread_borders(float mine, float (&b)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) b[j] = bits_f32(__builtin_amdgcn_readlane(f32_bits(mine), j));
}

// Work distribution shared by the streaming kernels.  A TILE is U*64 consecutive full groups; wave w of
// the launch owns tiles w, w + nwaves, w + 2*nwaves, ... so that at any moment the resident waves read
// one contiguous window of memory.  Whatever does not fill a tile (fewer than U*64 full groups plus the
// ragged last group) is the TAIL, done element-wise by the last wave.
// wave / nwaves / ntiles are wave-uniform and kept in SGPRs (readfirstlane), so the tile loop is a
// scalar loop and every address is "scalar base + lane offset".
struct Span {
    size_t wave, nwaves, ntiles, tail_g0, ngroups;
    size_t t0, t_end, stride;     // this wave's tiles: t0, t0 + stride, ... < t_end
    bool tail_owner;              // the one wave that also does the ragged tail
    int lane;
};

// Two ways of handing tiles to waves (host side: launch_shape()):
//   chunk == 0  RESIDENT: the grid is one resident generation of waves and wave w takes tiles w, w + W, w + 2W, ...
//               (the whole chip sweeps one contiguous window; best while the tensor is a few tiles per wave)
//   chunk == T  CHUNKED: block b owns the contiguous tiles [b*WPB*T, (b+1)*WPB*T) and its wave j takes
//               b*WPB*T + j, + WPB, ... (T tiles); the grid has as many blocks as that needs, far more than fit, and the
//               hardware dispatcher hands the next block to whichever CU frees up first -- dynamic load balancing at
//               block granularity for free (device-scope atomics cost ~10 ns per claim on this part), which is what a
//               large tensor needs: CUs/XCDs do not progress at the same rate and a static split waits for the slowest.
// `slack`: full groups that must remain after the last tile (the wide-code backward over-reads a few bytes)
template <int U, int WPB = kWavesPerBlock> __device__ __forceinline__ Span make_span(size_t n, int chunk, size_t slack = 0) {
    Span s;
    s.lane = threadIdx.x & (kWave - 1);
    const size_t wib = static_cast<size_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
    s.nwaves = static_cast<size_t>(gridDim.x) * WPB;
    s.wave = static_cast<size_t>(blockIdx.x) * WPB + wib;
    const size_t full = n >> 3;
    s.ntiles = (full > slack ? full - slack : 0) / (static_cast<size_t>(U) * kWave);
    s.tail_g0 = s.ntiles * (static_cast<size_t>(U) * kWave);
    s.ngroups = (n + 7) >> 3;
    if (chunk > 0) {
        const size_t per_block = static_cast<size_t>(WPB) * static_cast<size_t>(chunk);
        const size_t b0 = static_cast<size_t>(blockIdx.x) * per_block;
        s.t0 = b0 + wib;
        s.t_end = b0 + per_block < s.ntiles ? b0 + per_block : s.ntiles;
        s.stride = WPB;
    } else {
        s.t0 = s.wave;
        s.t_end = s.ntiles;
        s.stride = s.nwaves;
    }
    s.tail_owner = s.wave == s.nwaves - 1;
    return s;
}

// Two-buffer software pipeline over the tiles of one wave: the loads of the next tile are issued
// before the current tile is processed, and the two register buffers alternate (no copies), so the
// only wait in front of process(tile i) is for loads issued a whole tile earlier.
//   load(t, buf)    issues the global loads of tile t into buf (no use of the data)
//   process(t, buf) consumes buf and stores the results of tile t
// Neither may contain an exec-divergent branch around a memory instruction (same s_waitcnt reason).
// The input pointers of these kernels are deliberately NOT __restrict__: outputs may alias inputs
// (in-place, as the reference op), and with noalias inputs LLVM sinks the prefetch loads below the
// stores of process(), right in front of their use, which undoes the pipeline.
//   init()          runs once, after the first two tiles' loads are in flight (table setup hides there)
//   EARLY           both buffers are requested before init() (the pattern-table forward only, see below)
template <typename Buf, bool EARLY = false, typename Init, typename Load, typename Process>
__device__ __forceinline__ void pipeline2(const Span &s, Init &&init, Load &&load, Process &&process) {
    Buf A, B;
    size_t t = s.t0;
    if (t >= s.t_end) {
        init();
        return;
    }
    // a prefetch past the wave's own tiles is redirected to ONE 16-byte piece shared by every wave of the launch (lane 0's
    // piece of the tensor's last tile, requested by all 64 lanes: one cache line, always hot, the data is dropped) --
    // not to a tile of its own (evicted by then: a real extra read, measured +1 us at 4096x4096), not past the chunk into
    // a neighbour's tiles, and not to a whole hot tile either (1-2 KiB of L2 -> L1 fill per wave for nothing: with one or
    // two tiles per wave that is as much fill traffic again as the real reads)
    const size_t hot = s.ntiles - 1;
    load(t, s.lane, A);
    if constexpr (EARLY) {
    // head: BOTH buffers are requested before init() (table build / LDS staging + barrier), so that the memory system
    // has two tiles per wave in flight while the block sets itself up (pattern-table forward 12.0 -> 11.2 us at
    // 4096x4096 bf16; bucketing the first tile by register search ahead of the barrier, or building the table with one
    // barrier, did not help: EXPERIMENTS.md section 6).  Prefetches are unconditional (redirected to `hot` past the end, the data
    // is then simply dropped): a load issued on only one path would make the s_waitcnt in front of process() count for
    // the shorter path and wait for the prefetch itself.
    size_t t1 = t + s.stride;
    load(t1 < s.t_end ? t1 : hot, t1 < s.t_end ? s.lane : 0, B);
    init();
    for (;;) {
        process(t, A);
        if (t1 >= s.t_end) break;
        const size_t t2 = t1 + s.stride;
        load(t2 < s.t_end ? t2 : hot, t2 < s.t_end ? s.lane : 0, A);
        process(t1, B);
        if (t2 >= s.t_end) break;
        t = t2;
        t1 = t2 + s.stride;
        load(t1 < s.t_end ? t1 : hot, t1 < s.t_end ? s.lane : 0, B);
    }
    } else {
    init();
    for (;;) {
        const size_t t1 = t + s.stride;
        load(t1 < s.t_end ? t1 : hot, t1 < s.t_end ? s.lane : 0, B);
        process(t, A);
        if (t1 >= s.t_end) break;
        const size_t t2 = t1 + s.stride;
        load(t2 < s.t_end ? t2 : hot, t2 < s.t_end ? s.lane : 0, A);
        process(t1, B);
        if (t2 >= s.t_end) break;
        t = t2;
    }
    }
}

// occupancy each forward instantiation is compiled for: 8 waves/SIMD (<= 64 VGPRs) where the table and
// the two prefetch buffers fit without spilling, 6 (<= 80) for 4-bit tables (15 border VGPRs), fp32
// groups (8 VGPRs per buffer) and mish (ocml log1p+exp+tanh).  The launcher sizes the grid from the occupancy the runtime reports.
template <int FN, int DT, int K, int U = 1> constexpr int forward_waves_per_simd() {
    if (U >= 4) return 4;                      // 2 x 4 raw groups per lane in flight: <= 128 VGPRs
    if (U == 2) return (K == 4 || DT == FEWBIT_F32 || FN == FEWBIT_MISH) ? 4 : 6;
    return (K == 4 || DT == FEWBIT_F32 || FN == FEWBIT_MISH) ? 6 : kWavesPerSimd;
}
// backward / 1-bit kernels: 8 waves per SIMD up to two groups per lane per stage, 6 (<= 80 VGPRs) beyond
template <int DT, int U> constexpr int stream_waves_per_simd() {
    return (U >= 4 || (U >= 2 && DT == FEWBIT_F32)) ? 6 : kWavesPerSimd;
}

// ------------------------------------------------------------------------------------------------
// Fused forward for 2^K-level tables, K in 1..4: activation + quantize + pack, software-pipelined:
// the loads of tile i+1 are in flight while tile i is bucketed, evaluated and stored.
// Replaces StepwiseKernel<Fn> + BinarySearch + DeflateWarpKernel (fewbit/cuda/codec.cu:489-504,
// :118-131, :142-165).
template <int FN, int DT, int K, int U>
__global__ __launch_bounds__(kBlock, (forward_waves_per_simd<FN, DT, K, U>())) void quantize_forward_kernel(const void *x, void *y,
                                                                  uint8_t *__restrict__ state, size_t n,
                                                                  const void *__restrict__ borders, float p0,
                                                                  float p1, int chunk) {
    constexpr int NB = (1 << K) - 1;
    constexpr bool kFast = (DT != FEWBIT_F32);
    constexpr bool kSplit = (DT == FEWBIT_F32) && (FEWBIT_F32_SPLIT != 0);     // fp32: contiguous split tiles
    constexpr bool kStreamY = (DT != FEWBIT_F32) || kSplit;
    typedef typename GroupIO<DT>::Raw Raw;
    const Span s = make_span<U>(n, chunk);

    float b[NB];
    const float mine = fetch_border<DT, NB>(borders);

    struct Buf { Raw r[U]; };
    pipeline2<Buf>(
        s, [&]() { spread_borders<NB>(mine, b); },
        [&](size_t t, int ln, Buf &buf) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (kSplit) {
                    const SplitF32::Raw r = SplitF32::load_raw(x, (t * U + u) * kWave + ln, ln);
                    buf.r[u].a = r.a;
                    buf.r[u].b = r.b;
                } else {
                    buf.r[u] = GroupIO<DT>::load_raw(x, (t * U + u) * kWave + ln);
                }
            }
        },
        [&](size_t t, const Buf &buf) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float v[8];
                GroupIO<DT>::unpack(buf.r[u], v);      // (the split layout unpacks the same way: v[0..3] | v[4..7] are its two halves)
                uint32_t w;
                float key[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) key[i] = Act<FN, kFast>::key(v[i], p0);      // x itself unless folded
                if constexpr (kSplit) {
                    const float ka[4] = {key[0], key[1], key[2], key[3]}, kb[4] = {key[4], key[5], key[6], key[7]};
                    w = split_halves_to_word<K>(pack_half<K>(ka, b), pack_half<K>(kb, b), s.lane);
                } else {
                    w = pack_group<K>(key, b);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = Act<FN, kFast>::eval(v[i], p0, p1);
                const size_t g = (t * U + u) * kWave + s.lane;
                // y: nontemporal for 16-bit dtypes (not read again here; keeps the remaining input resident in L2 /
                // Infinity Cache).  fp32 groups are stored as two 16 B pieces at a 32 B lane stride, i.e. each store
                // instruction leaves holes that only L2 write-combining fills -- nontemporal there costs 4 us per pass.
                // state: plain store -- it is what backward reads, and a backward that follows closely finds it
                // cached (4096x4096 bf16 step 26.5 -> 25.6 us); when backward runs much later it makes no difference.
                if constexpr (kSplit) SplitF32::store<true>(y, g, s.lane, v);
                else GroupIO<DT>::template store<kStreamY>(y, g, v);
                store_state_quad<K, false>(state, g, s.lane, w);
            }
        });

    // ---- tail: per-group and per-element guards, last wave only
    if (!s.tail_owner) return;
    for (size_t g = s.tail_g0 + s.lane; g < s.ngroups; g += kWave) {
        const size_t e0 = g << 3;
        uint32_t w = 0;
        for (int i = 0; i < 8; ++i) {
            if (e0 + i < n) {
                float xv = Elem<DT>::load(x, e0 + i);
                w |= count_below<NB>(b, Act<FN, kFast>::key(xv, p0)) << (K * i);
                Elem<DT>::store(y, e0 + i, Act<FN, kFast>::eval(xv, p0, p1));
            }
        }
        store_state<K>(state, g, w);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused forward for 16-bit dtypes with the border search replaced by a table lookup in LDS.
//
// A 16-bit input has 65 536 possible bit patterns, so "code of x" is a 64 KiB byte table indexed by the raw pattern.
// It is built per block in three cheap passes (the code is a step function of the pattern: ascending over
// 0x0000..0x7fff, descending over 0x8000..0xffff):
//   1. every thread fills its 64-pattern chunk with the code of the chunk's first pattern (float predicate,
//      identical to the search kernel's; NaN chunks get `nborders`, torch.searchsorted's rule);
//   2. wave j corrects, inside the single chunk border j falls into, the patterns that lie beyond it (+1 in the
//      positive half, -1 in the negative half) with LDS atomics, while the last wave fixes the NaN patterns that
//      share a chunk with +-inf.
// After that bucketing costs one ds_read_u8 and a shift-or per element instead of 2^K-1+K narrow-class VALU
// instructions -- independent of K -- and runs on the otherwise idle LDS unit, which takes the forward from
// VALU-bound back to memory-bound.  Blocks are 1024 threads (16 waves) so that two of them (2 x 64 KiB of LDS) fill
// a CU with 32 waves.
constexpr int kLutBlock = 1024;   // threads per block of the pattern-table kernels
constexpr int kLutWaves = kLutBlock / kWave;
// two blocks per CU (2 x 64 KiB of LDS): 1024 threads -> 8 waves per SIMD (<= 64 VGPRs), 512 threads -> 4 (<= 128)
template <int BLOCK> constexpr int lut_waves_per_simd() { return 2 * BLOCK / 256; }

template <int DT> __device__ __forceinline__ float value_of_pattern(uint32_t r) {
    if constexpr (DT == FEWBIT_BF16) return bits_f32(r << 16);
    else return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(r)));
}

template <int FN, int DT, int K, int U, int BLOCK = kLutBlock>
__global__ __launch_bounds__(BLOCK, (lut_waves_per_simd<BLOCK>())) void quantize_forward_lut_kernel(const void *x, void *y, uint8_t *state,
                                                                            size_t n, const void *borders,
                                                                            int nborders, float p0, float p1, int chunk) {
    constexpr int kLutBlock = BLOCK, kLutWaves = BLOCK / kWave;        // (shadow the file-level defaults)
    static_assert(DT != FEWBIT_F32, "the pattern table exists for 16-bit dtypes only");
    constexpr int NBMAX = (1 << K) - 1;
    constexpr uint32_t kInf = (DT == FEWBIT_BF16) ? 0x7f80u : 0x7c00u;
    typedef typename GroupIO<DT>::Raw Raw;
    __shared__ __attribute__((aligned(16))) uint8_t lut[65536];
    const Span s = make_span<U, kLutWaves>(n, chunk);

    // the table's global loads go out FIRST (lane j fetches border j, as a float and as a raw pattern): the build then
    // waits only for them, not for the first tile of x that pipeline2 issues right after
    float mine = __builtin_inff();
    uint32_t mine_raw = 0;
    if (s.lane < nborders) {
        mine = Elem<DT>::load(borders, s.lane);
        mine_raw = static_cast<const uint16_t *>(borders)[s.lane];
    }

    struct Buf { Raw r[U]; };
    auto build = [&]() {
        // wave-uniform copy of the table (padding +inf never satisfies !(b >= x) for a number)
        float b[NBMAX];
#pragma unroll
        for (int j = 0; j < NBMAX; ++j) b[j] = bits_f32(__builtin_amdgcn_readlane(f32_bits(mine), j));
        // pass 1: chunks of 64 patterns, one per thread (1024-thread blocks); a chunk that starts on a NaN pattern is all
        // NaN -> nborders
#pragma unroll
        for (uint32_t c = threadIdx.x; c < 1024u; c += kLutBlock) {
            const uint32_t r0 = c * 64u;
            const uint32_t c_first = ((r0 & 0x7fffu) > kInf) ? static_cast<uint32_t>(nborders)
                                                            : count_below<NBMAX>(b, value_of_pattern<DT>(r0));
            const uint32_t c0 = c_first * 0x01010101u;
            u32x4 fill = {c0, c0, c0, c0};
            u32x4 *dst = reinterpret_cast<u32x4 *>(lut + r0);
            dst[0] = fill; dst[1] = fill; dst[2] = fill; dst[3] = fill;
        }
        __syncthreads();
        // pass 2: wave j corrects border j's chunk, all borders at once: lane l owns the l-th pattern past the border
        // and adds/subtracts 1 in its byte with an LDS dword atomic (no carry: codes stay within 0..15).  NaN patterns
        // are never touched here; the last wave rewrites the 63 NaN patterns that share a chunk with +-inf.
        const int wv = threadIdx.x >> 6;
        uint32_t *lut32 = reinterpret_cast<uint32_t *>(lut);
        for (int j = wv; j < nborders; j += kLutWaves) {                   // wave-uniform; one round with 16 waves and K <= 4
            const uint32_t bits = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine_raw), j));
            const uint32_t mag = bits & 0x7fffu;
            const bool neg = (bits >> 15) != 0 && mag != 0;               // -0 behaves as +0
            if (mag <= kInf) {                                              // a NaN border is counted for every x already
                // first pattern whose predicate differs from that of its chunk's first pattern
                const uint32_t first = neg ? 0x8000u + mag : mag + 1u;
                const uint32_t r = first + s.lane;
                if ((first & 63u) != 0 && r < ((first | 63u) + 1u) && (r & 0x7fffu) <= kInf) {
                    const uint32_t one = 1u << (8u * (r & 3u));
                    if (neg) __hip_atomic_fetch_sub(&lut32[r >> 2], one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_fetch_add(&lut32[r >> 2], one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (wv == kLutWaves - 1 && s.lane > 0) {                           // kInf is chunk aligned for bf16 and fp16
            lut[kInf + s.lane] = static_cast<uint8_t>(nborders);
            lut[0x8000u + kInf + s.lane] = static_cast<uint8_t>(nborders);
        }
        __syncthreads();
    };

    pipeline2<Buf, true>(
        s, build,
        [&](size_t t, int ln, Buf &buf) {
#pragma unroll
            for (int u = 0; u < U; ++u) buf.r[u] = GroupIO<DT>::load_raw(x, (t * U + u) * kWave + ln);
        },
        [&](size_t t, const Buf &buf) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t w = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t d = buf.r[u].q[i];
                    w |= static_cast<uint32_t>(lut[d & 0xffffu]) << (K * 2 * i);
                    w |= static_cast<uint32_t>(lut[d >> 16]) << (K * (2 * i + 1));
                }
                float v[8];
                GroupIO<DT>::unpack(buf.r[u], v);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = Act<FN, true>::eval(v[i], p0, p1);
                const size_t g = (t * U + u) * kWave + s.lane;
                GroupIO<DT>::template store<true>(y, g, v);
                store_state_quad<K, false>(state, g, s.lane, w);      // plain store: a backward that follows closely finds it cached
            }
        });

    // ---- tail: element-wise through the same table, last wave only
    if (!s.tail_owner) return;
    for (size_t g = s.tail_g0 + s.lane; g < s.ngroups; g += kWave) {
        const size_t e0 = g << 3;
        uint32_t w = 0;
        for (int i = 0; i < 8; ++i) {
            if (e0 + i < n) {
                const uint32_t r = static_cast<const uint16_t *>(x)[e0 + i];
                w |= static_cast<uint32_t>(lut[r]) << (K * i);
                Elem<DT>::store(y, e0 + i, Act<FN, true>::eval(value_of_pattern<DT>(r), p0, p1));
            }
        }
        store_state<K>(state, g, w);
    }
}

// ------------------------------------------------------------------------------------------------
// Wide tables (17..256 levels, 5..8 bits per code) and tables that do not fill their bit width (3, 5..7, 9..15
// levels): the same streaming structure with the
// bit width as a run-time (wave-uniform) value, so that one instantiation per functor/dtype serves all four widths.

// code of `key` against the padded border tree in LDS: fixed nbits-step descent, NaN -> nborders
__device__ __forceinline__ uint32_t tree_code(const float *sb, float key, int nbits, uint32_t nborders) {
    uint32_t pos = 0;
    for (uint32_t step = 1u << (nbits - 1); step != 0; step >>= 1) pos += !(sb[pos + step - 1] >= key) ? step : 0u;
    return min(pos, nborders);
}

// element-wise tail shared by the wide forward kernels (last wave only)
template <int FN, int DT, typename Code>
__device__ __forceinline__ void wide_forward_tail(const Span &s, const void *x, void *y, uint8_t *state, size_t n, int nbits,
                                                  float p0, float p1, Code &&code_of) {
    constexpr bool kFast = (DT != FEWBIT_F32);
    for (size_t g = s.tail_g0 + s.lane; g < s.ngroups; g += kWave) {
        const size_t e0 = g << 3;
        uint64_t w = 0;
        for (int i = 0; i < 8; ++i) {
            if (e0 + i < n) {
                w |= static_cast<uint64_t>(code_of(e0 + i)) << (nbits * i);
                Elem<DT>::store(y, e0 + i, Act<FN, kFast>::eval(Elem<DT>::load(x, e0 + i), p0, p1));
            }
        }
        uint8_t *p = state + static_cast<size_t>(nbits) * g;
        for (int j = 0; j < nbits; ++j) p[j] = static_cast<uint8_t>(w >> (8 * j));
    }
}

// forward, borders searched in LDS (fp32 tensors; 16-bit tensors below the pattern-table threshold)
template <int FN, int DT>
__global__ __launch_bounds__(kBlock, 6) void quantize_forward_wide_kernel(const void *x, void *y, uint8_t *state, size_t n,
                                                                          const void *borders, int nborders, int nbits,
                                                                          float p0, float p1, int chunk) {
    constexpr bool kFast = (DT != FEWBIT_F32);
    typedef typename GroupIO<DT>::Raw Raw;
    __shared__ float sb[256];
    const Span s = make_span<1>(n, chunk);
    struct Buf { Raw r; };
    pipeline2<Buf>(
        s,
        [&]() {
            sb[threadIdx.x] = static_cast<int>(threadIdx.x) < nborders ? Elem<DT>::load(borders, threadIdx.x) : __builtin_inff();
            __syncthreads();
        },
        [&](size_t t, int ln, Buf &buf) { buf.r = GroupIO<DT>::load_raw(x, t * kWave + ln); },
        [&](size_t t, const Buf &buf) {
            float v[8];
            GroupIO<DT>::unpack(buf.r, v);
            uint32_t pos[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pos[i] = 0;
            for (uint32_t step = 1u << (nbits - 1); step != 0; step >>= 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) pos[i] += !(sb[pos[i] + step - 1] >= Act<FN, kFast>::key(v[i], p0)) ? step : 0u;
            }
            // two 32-bit halves (4 codes of <= 8 bits each) and ONE 64-bit shift instead of eight
            uint32_t wlo = 0, whi = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wlo |= min(pos[i], static_cast<uint32_t>(nborders)) << (nbits * i);
                whi |= min(pos[4 + i], static_cast<uint32_t>(nborders)) << (nbits * i);
            }
            const uint64_t w = static_cast<uint64_t>(wlo) | (static_cast<uint64_t>(whi) << (4 * nbits));
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = Act<FN, kFast>::eval(v[i], p0, p1);
            const size_t g = t * kWave + s.lane;
            GroupIO<DT>::template store<kFast>(y, g, v);
            store_state_wide(state, g, nbits, w);
        });
    if (!s.tail_owner) return;
    wide_forward_tail<FN, DT>(s, x, y, state, n, nbits, p0, p1, [&](size_t e) {
        return tree_code(sb, Act<FN, kFast>::key(Elem<DT>::load(x, e), p0), nbits, static_cast<uint32_t>(nborders));
    });
}

// forward for 16-bit tensors through the 64 KiB pattern table, any number of borders up to 255: the borders are
// staged in LDS (float for the tree descent of pass 1, raw pattern for the fix-up of pass 2, which loops over them
// 16 waves at a time); everything else as in quantize_forward_lut_kernel.
template <int FN, int DT>
__global__ __launch_bounds__(kLutBlock, (lut_waves_per_simd<kLutBlock>())) void quantize_forward_lut_wide_kernel(const void *x, void *y,
                                                                                 uint8_t *state, size_t n,
                                                                                 const void *borders, int nborders,
                                                                                 int nbits, float p0, float p1, int chunk) {
    static_assert(DT != FEWBIT_F32, "the pattern table exists for 16-bit dtypes only");
    constexpr uint32_t kInf = (DT == FEWBIT_BF16) ? 0x7f80u : 0x7c00u;
    typedef typename GroupIO<DT>::Raw Raw;
    __shared__ __attribute__((aligned(16))) uint8_t lut[65536];
    __shared__ float sb[256];
    __shared__ uint16_t sr[256];
    const Span s = make_span<1, kLutWaves>(n, chunk);
    float mine = __builtin_inff();
    uint32_t mine_raw = 0;
    if (static_cast<int>(threadIdx.x) < nborders) {
        mine = Elem<DT>::load(borders, threadIdx.x);
        mine_raw = static_cast<const uint16_t *>(borders)[threadIdx.x];
    }
    struct Buf { Raw r; };
    auto build = [&]() {
        if (threadIdx.x < 256) {
            sb[threadIdx.x] = mine;
            sr[threadIdx.x] = static_cast<uint16_t>(mine_raw);
        }
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < 1024u; c += kLutBlock) {
            const uint32_t r0 = c * 64u;
            const uint32_t c_first = ((r0 & 0x7fffu) > kInf) ? static_cast<uint32_t>(nborders)
                                                            : tree_code(sb, value_of_pattern<DT>(r0), nbits, static_cast<uint32_t>(nborders));
            const uint32_t c0 = c_first * 0x01010101u;
            u32x4 fill = {c0, c0, c0, c0};
            u32x4 *dst = reinterpret_cast<u32x4 *>(lut + r0);
            dst[0] = fill; dst[1] = fill; dst[2] = fill; dst[3] = fill;
        }
        __syncthreads();
        uint32_t *lut32 = reinterpret_cast<uint32_t *>(lut);
        for (int j = threadIdx.x >> 6; j < nborders; j += kLutWaves) {     // wave-uniform loop
            const uint32_t bits = sr[j];
            const uint32_t mag = bits & 0x7fffu;
            const bool neg = (bits >> 15) != 0 && mag != 0;
            if (mag <= kInf) {
                const uint32_t first = neg ? 0x8000u + mag : mag + 1u;
                const uint32_t r = first + s.lane;
                if ((first & 63u) != 0 && r < ((first | 63u) + 1u) && (r & 0x7fffu) <= kInf) {
                    const uint32_t one = 1u << (8u * (r & 3u));
                    if (neg) __hip_atomic_fetch_sub(&lut32[r >> 2], one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_fetch_add(&lut32[r >> 2], one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __syncthreads();      // (the NaN rewrite below must follow every fix-up of the +-inf chunks)
        if ((threadIdx.x >> 6) == kLutWaves - 1 && s.lane > 0) {
            lut[kInf + s.lane] = static_cast<uint8_t>(nborders);
            lut[0x8000u + kInf + s.lane] = static_cast<uint8_t>(nborders);
        }
        __syncthreads();
    };
    pipeline2<Buf, true>(
        s, build, [&](size_t t, int ln, Buf &buf) { buf.r = GroupIO<DT>::load_raw(x, t * kWave + ln); },
        [&](size_t t, const Buf &buf) {
            uint32_t wlo = 0, whi = 0;        // elements 0..3 | 4..7: 32-bit halves, one 64-bit shift to join them
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t d0 = buf.r.q[i], d1 = buf.r.q[2 + i];
                wlo |= static_cast<uint32_t>(lut[d0 & 0xffffu]) << (nbits * (2 * i));
                wlo |= static_cast<uint32_t>(lut[d0 >> 16]) << (nbits * (2 * i + 1));
                whi |= static_cast<uint32_t>(lut[d1 & 0xffffu]) << (nbits * (2 * i));
                whi |= static_cast<uint32_t>(lut[d1 >> 16]) << (nbits * (2 * i + 1));
            }
            const uint64_t w = static_cast<uint64_t>(wlo) | (static_cast<uint64_t>(whi) << (4 * nbits));
            float v[8];
            GroupIO<DT>::unpack(buf.r, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = Act<FN, true>::eval(v[i], p0, p1);
            const size_t g = t * kWave + s.lane;
            GroupIO<DT>::template store<true>(y, g, v);
            store_state_wide(state, g, nbits, w);
        });
    if (!s.tail_owner) return;
    wide_forward_tail<FN, DT>(s, x, y, state, n, nbits, p0, p1,
                              [&](size_t e) { return static_cast<uint32_t>(lut[static_cast<const uint16_t *>(x)[e]]); });
}

// backward for wide tables: one unaligned 8-byte state load per group (one group of slack kept before the end)
template <int DT>
__global__ __launch_bounds__(kBlock, kWavesPerSimd) void quantize_backward_wide_kernel(const void *gy, const uint8_t *state,
                                                                                  void *gx, size_t n, const void *levels,
                                                                                  int nlevels, int nbits, int chunk) {
    typedef typename GroupIO<DT>::Raw Raw;
    __shared__ float lut[256];
    const Span s = make_span<1>(n, chunk, 1);
    const uint32_t mask = (1u << nbits) - 1u;
    struct Buf { Raw r; uint64_t w; };
    pipeline2<Buf>(
        s,
        [&]() {
            lut[threadIdx.x] = static_cast<int>(threadIdx.x) < nlevels ? Elem<DT>::load(levels, threadIdx.x) : 0.0f;
            __syncthreads();
        },
        [&](size_t t, int ln, Buf &buf) {
            buf.r = GroupIO<DT>::load_raw(gy, t * kWave + ln);
            buf.w = load_state_wide(state, t * kWave + ln, nbits);
        },
        [&](size_t t, const Buf &buf) {
            float v[8];
            GroupIO<DT>::unpack(buf.r, v);
            const uint32_t wlo = static_cast<uint32_t>(buf.w), whi = static_cast<uint32_t>(buf.w >> (4 * nbits));   // codes 0..3 | 4..7
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = lut[(wlo >> (nbits * i)) & mask] * v[i];
                v[4 + i] = lut[(whi >> (nbits * i)) & mask] * v[4 + i];
            }
            GroupIO<DT>::template store<(DT != FEWBIT_F32)>(gx, t * kWave + s.lane, v);
        });
    if (!s.tail_owner) return;
    for (size_t g = s.tail_g0 + s.lane; g < s.ngroups; g += kWave) {
        const size_t e0 = g << 3;
        const uint8_t *p = state + static_cast<size_t>(nbits) * g;
        uint64_t w = 0;
        for (int j = 0; j < nbits; ++j) w |= static_cast<uint64_t>(p[j]) << (8 * j);
        for (int i = 0; i < 8; ++i)
            if (e0 + i < n)
                Elem<DT>::store(gx, e0 + i, mul_f32(lut[static_cast<uint32_t>(w >> (nbits * i)) & mask], Elem<DT>::load(gy, e0 + i)));
    }
}

// ------------------------------------------------------------------------------------------------
// Fused backward: unpack + level gather (LDS) + multiply, same software pipeline.
// Replaces StepwiseBackwardKernel + InflateWarpKernel (fewbit/cuda/codec.cu:655-663, :184-203).
template <int DT, int K, int U>
__global__ __launch_bounds__(kBlock, (stream_waves_per_simd<DT, U>())) void quantize_backward_kernel(const void *gy,
                                                                   const uint8_t *state, void *gx,
                                                                   size_t n, const void *__restrict__ levels,
                                                                   int nlevels, int chunk) {
    constexpr int NL = 1 << K;
    constexpr uint32_t kMask = NL - 1;
    typedef typename GroupIO<DT>::Raw Raw;
    __shared__ float lut[NL];
    const Span s = make_span<U>(n, chunk);

    // the level fetch goes out before the first tiles: vmcnt counts in order, so waiting for it in init() does not
    // wait for the tiles as well
    constexpr bool kSplit = (DT == FEWBIT_F32) && (FEWBIT_F32_SPLIT != 0);
    float mine = 0.0f;
    if (threadIdx.x < NL && static_cast<int>(threadIdx.x) < nlevels) mine = Elem<DT>::load(levels, threadIdx.x);
    struct Buf { Raw r[U]; uint32_t w[U]; };
    pipeline2<Buf>(
        s,
        [&]() {  // every wave of the block gets here exactly once, so the barrier is safe
            if (threadIdx.x < NL) lut[threadIdx.x] = mine;
            __syncthreads();
        },
        [&](size_t t, int ln, Buf &buf) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (kSplit) {
                    const SplitF32::Raw r = SplitF32::load_raw(gy, (t * U + u) * kWave + ln, ln);
                    buf.r[u].a = r.a;
                    buf.r[u].b = r.b;
                } else {
                    buf.r[u] = GroupIO<DT>::load_raw(gy, (t * U + u) * kWave + ln);
                }
                buf.w[u] = load_state_quad_raw<K>(state, (t * U + u) * kWave + ln, ln);
            }
        },
        [&](size_t t, const Buf &buf) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float v[8];
                GroupIO<DT>::unpack(buf.r[u], v);
                const uint32_t w = load_state_quad_fix<K>(buf.w[u], s.lane);
                if constexpr (kSplit) {      // v[0..3] / v[4..7] are halves of two different groups (SplitF32)
                    uint32_t cA, cB;
                    split_word_to_halves<K>(w, s.lane, cA, cB);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = lut[(cA >> (K * i)) & kMask] * v[i];
                        v[4 + i] = lut[(cB >> (K * i)) & kMask] * v[4 + i];
                    }
                    SplitF32::store<true>(gx, (t * U + u) * kWave + s.lane, s.lane, v);
                    continue;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = lut[(w >> (K * i)) & kMask] * v[i];
                // gx of a 16-bit backward: nontemporal, like y in the forward -- a write-allocated gx pushes what is still to
                // be read out of L2 / Infinity Cache and is written back during the NEXT kernel (cache-cold backward at
                // 4096x4096 bf16 15.3 -> 13.9 us, 2^26 elements forward+backward 100.6 -> 94.1 us; RoBERTa-base step
                // unchanged).  fp32 in the split layout (above) stores whole lines and is nontemporal too.
                GroupIO<DT>::template store<(DT != FEWBIT_F32)>(gx, (t * U + u) * kWave + s.lane, v);
            }
        });

    if (!s.tail_owner) return;
    for (size_t g = s.tail_g0 + s.lane; g < s.ngroups; g += kWave) {
        const size_t e0 = g << 3;
        const uint32_t w = load_state<K>(state, g);
        for (int i = 0; i < 8; ++i)
            if (e0 + i < n) Elem<DT>::store(gx, e0 + i, mul_f32(lut[(w >> (K * i)) & kMask], Elem<DT>::load(gy, e0 + i)));
    }
}

// ------------------------------------------------------------------------------------------------
// 1-bit family (ReLU & co): exact derivative, one bit per element.
// Replaces the eight <Name>Kernel / <Name>BackwardKernel pairs, fewbit/cuda/codec.cu:298-487.
template <int FN, int DT, int U>
__global__ __launch_bounds__(kBlock, (stream_waves_per_simd<DT, U>())) void stepwise1_forward_kernel(const void *x, void *y,
                                                                   uint8_t *__restrict__ state, size_t n, float p0,
                                                                   float p1, int chunk) {
    typedef typename GroupIO<DT>::Raw Raw;
    const Span s = make_span<U>(n, chunk);
    constexpr bool kSplit = (DT == FEWBIT_F32) && (FEWBIT_F32_SPLIT != 0);     // see SplitF32
    struct Buf { Raw r[U]; };
    pipeline2<Buf>(
        s, []() {},
        [&](size_t t, int ln, Buf &buf) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (kSplit) {
                    const SplitF32::Raw r = SplitF32::load_raw(x, (t * U + u) * kWave + ln, ln);
                    buf.r[u].a = r.a;
                    buf.r[u].b = r.b;
                } else {
                    buf.r[u] = GroupIO<DT>::load_raw(x, (t * U + u) * kWave + ln);
                }
            }
        },
        [&](size_t t, const Buf &buf) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float v[8];
                GroupIO<DT>::unpack(buf.r[u], v);
                uint32_t w = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    uint32_t bit;
                    v[i] = Step1<FN>::eval(v[i], p0, p1, bit);
                    w |= bit << i;
                }
                const size_t g = (t * U + u) * kWave + s.lane;
                if constexpr (kSplit) {      // bits 0..3 / 4..7 of w belong to halves of two different groups
                    w = split_halves_to_word<1>(w & 15u, w >> 4, s.lane);
                    SplitF32::store<true>(y, g, s.lane, v);
                } else {
                    GroupIO<DT>::template store<(DT != FEWBIT_F32)>(y, g, v);
                }
                store_state_quad<1, false>(state, g, s.lane, w);
            }
        });
    if (!s.tail_owner) return;
    for (size_t g = s.tail_g0 + s.lane; g < s.ngroups; g += kWave) {
        const size_t e0 = g << 3;
        uint32_t w = 0;
        for (int i = 0; i < 8; ++i) {
            if (e0 + i < n) {
                uint32_t bit;
                Elem<DT>::store(y, e0 + i, Step1<FN>::eval(Elem<DT>::load(x, e0 + i), p0, p1, bit));
                w |= bit << i;
            }
        }
        store_state<1>(state, g, w);
    }
}

template <int DT, int U>
__global__ __launch_bounds__(kBlock, (stream_waves_per_simd<DT, U>())) void stepwise1_backward_kernel(const void *gy,
                                                                    const uint8_t *state, void *gx,
                                                                    size_t n, float m0, float m1, int chunk) {
    typedef typename GroupIO<DT>::Raw Raw;
    const Span s = make_span<U>(n, chunk);
    constexpr bool kSplit = (DT == FEWBIT_F32) && (FEWBIT_F32_SPLIT != 0);
    struct Buf { Raw r[U]; uint32_t w[U]; };
    pipeline2<Buf>(
        s, []() {},
        [&](size_t t, int ln, Buf &buf) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (kSplit) {
                    const SplitF32::Raw r = SplitF32::load_raw(gy, (t * U + u) * kWave + ln, ln);
                    buf.r[u].a = r.a;
                    buf.r[u].b = r.b;
                } else {
                    buf.r[u] = GroupIO<DT>::load_raw(gy, (t * U + u) * kWave + ln);
                }
                buf.w[u] = load_state_quad_raw<1>(state, (t * U + u) * kWave + ln, ln);
            }
        },
        [&](size_t t, const Buf &buf) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float v[8];
                GroupIO<DT>::unpack(buf.r[u], v);
                uint32_t w = load_state_quad_fix<1>(buf.w[u], s.lane);
                if constexpr (kSplit) {
                    uint32_t cA, cB;
                    split_word_to_halves<1>(w, s.lane, cA, cB);
                    w = (cA & 15u) | (cB << 4);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (((w >> i) & 1u) ? m1 : m0) * v[i];
                if constexpr (kSplit) SplitF32::store<true>(gx, (t * U + u) * kWave + s.lane, s.lane, v);
                else GroupIO<DT>::template store<(DT != FEWBIT_F32)>(gx, (t * U + u) * kWave + s.lane, v);
            }
        });
    if (!s.tail_owner) return;
    for (size_t g = s.tail_g0 + s.lane; g < s.ngroups; g += kWave) {
        const size_t e0 = g << 3;
        const uint32_t w = load_state<1>(state, g);
        for (int i = 0; i < 8; ++i)
            if (e0 + i < n) Elem<DT>::store(gx, e0 + i, mul_f32(((w >> i) & 1u) ? m1 : m0, Elem<DT>::load(gy, e0 + i)));
    }
}

#if !defined(FEWBIT_TU) || FEWBIT_TU == 0      // (not templates: one unit of a split build defines them)
// ------------------------------------------------------------------------------------------------
// Stand-alone codec (test seam for the layout): int32 codes <-> packed bytes, width 1..8.
// Replaces DeflateBlockKernel / InflateBlockKernel (fewbit/cuda/codec.cu:166-182, :205-220).
__global__ __launch_bounds__(kBlock) void pack_codes_kernel(const int32_t *__restrict__ codes,
                                                            uint8_t *__restrict__ state, size_t n, int nbits) {
    const size_t g = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    const size_t e0 = g << 3;
    if (e0 >= n) return;
    const uint64_t mask = (1ull << nbits) - 1ull;
    uint64_t w = 0;
    for (int i = 0; i < 8; ++i)
        if (e0 + i < n) w |= (static_cast<uint64_t>(static_cast<uint32_t>(codes[e0 + i])) & mask) << (nbits * i);
    uint8_t *p = state + static_cast<size_t>(nbits) * g;
    for (int j = 0; j < nbits; ++j) p[j] = static_cast<uint8_t>(w >> (8 * j));
}

__global__ __launch_bounds__(kBlock) void unpack_codes_kernel(const uint8_t *__restrict__ state,
                                                              int32_t *__restrict__ codes, size_t n, int nbits) {
    const size_t g = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
    const size_t e0 = g << 3;
    if (e0 >= n) return;
    const uint8_t *p = state + static_cast<size_t>(nbits) * g;
    uint64_t w = 0;
    for (int j = 0; j < nbits; ++j) w |= static_cast<uint64_t>(p[j]) << (8 * j);
    const uint64_t mask = (1ull << nbits) - 1ull;
    for (int i = 0; i < 8; ++i)
        if (e0 + i < n) codes[e0 + i] = static_cast<int32_t>((w >> (nbits * i)) & mask);
}
#endif

// ================================================================================================
// host side: argument checks, dispatch, launch
// ================================================================================================
// Split build (Makefile): the forward functors -- nine tenths of the compile time -- are compiled in parallel as units
// -DFEWBIT_TU=1..FEWBIT_TU_COUNT (three functors each, only dispatch_forward_dtype<FN> instantiated), everything else is unit
// -DFEWBIT_TU=0; without FEWBIT_TU one unit holds everything (`make variant`).  What the units share has external (hidden)
// linkage and is defined in the core unit; all the rest below is per unit.
#if defined(FEWBIT_TU) && FEWBIT_TU != 0
#define FEWBIT_CORE_TU 0
#else
#define FEWBIT_CORE_TU 1
#endif
#define FEWBIT_TU_COUNT 5
#define FEWBIT_HIDDEN __attribute__((visibility("hidden")))

// Run-time tuning.  Every key is -1 ("built-in policy") unless its environment variable is set when the library makes
// its first launch, or fewbit_hip_tune() sets it later (measurement scripts sweep shapes inside one process that way).
enum TuneKey {
    T_WAVES_PER_CU,        // cap on the resident waves per CU a streaming kernel's grid is sized for
    T_CHUNK,               // search / backward / 1-bit kernels: 0 = resident shape, T = chunked with T tiles per wave
    T_LUT_CHUNK,           // the same for the pattern-table forward
    T_LUT_BLOCKS_PER_CU,   // cap on resident pattern-table blocks per CU (at most 2 fit: 64 KiB of LDS each)
    T_LUT_MIN,             // smallest tensor (elements) that takes the pattern-table forward; 0 = always
    T_U_FWD, T_U_BWD, T_U_STEP1,     // groups per lane per pipeline stage (1 or 2), where the kernel has both
    T_COUNT
};
struct TuneSpec { const char *key, *env; };
constexpr TuneSpec kTuneSpec[T_COUNT] = {
    {"waves_per_cu", "FEWBIT_HIP_WAVES_PER_CU"}, {"chunk", "FEWBIT_HIP_CHUNK"}, {"lut_chunk", "FEWBIT_HIP_LUT_CHUNK"},
    {"lut_blocks_per_cu", "FEWBIT_HIP_LUT_BLOCKS_PER_CU"}, {"lut_min", "FEWBIT_HIP_LUT_MIN"},
    {"u_fwd", "FEWBIT_HIP_U_FWD"}, {"u_bwd", "FEWBIT_HIP_U_BWD"}, {"u_step1", "FEWBIT_HIP_U_STEP1"},
};

// What a call launched (or would launch: fewbit_hip_describe_*): kernel instantiation and launch shape.
struct Plan {
    char kernel[128];
    unsigned blocks;
    int threads, chunk, u, k, blocks_per_cu;
};

extern FEWBIT_HIDDEN thread_local char g_last_error[256];
extern FEWBIT_HIDDEN std::atomic<long long> g_tune[T_COUNT];
FEWBIT_HIDDEN int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
FEWBIT_HIDDEN void tune_init();
FEWBIT_HIDDEN int sketch_tune(const char *key, long long value, bool *known);      // fewbit_sketch.hip

#if FEWBIT_CORE_TU
thread_local char g_last_error[256] = "";
std::atomic<long long> g_tune[T_COUNT];

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof g_last_error, fmt, ap);
    va_end(ap);
    return code;
}

void tune_init() {
    static std::once_flag once;
    std::call_once(once, [] {
        for (int i = 0; i < T_COUNT; ++i) {
            const char *e = getenv(kTuneSpec[i].env);
            g_tune[i].store(e && *e ? atoll(e) : -1ll, std::memory_order_relaxed);
        }
    });
}
#endif

namespace {

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FEWBIT_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return FEWBIT_OK;
}


// FEWBIT_HIP_VALIDATE=1 (debug aid, off by default: two runtime queries per pointer): every pointer handed to the
// C-ABI must be device memory and [p, p + bytes) must lie inside its allocation -- a host pointer or a state buffer
// sized with the wrong bit width then fails with FEWBIT_ERR_INVALID_ARGUMENT instead of a fault on the GPU.
bool validate_enabled() {
    static const bool on = [] {
        const char *e = getenv("FEWBIT_HIP_VALIDATE");
        return e && atoi(e) != 0;
    }();
    return on;
}

int validate_range(const char *what, const void *p, size_t bytes) {
    if (!validate_enabled() || bytes == 0) return FEWBIT_OK;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(FEWBIT_ERR_INVALID_ARGUMENT, "%s (%p) is not a pointer known to the HIP runtime", what, p);
    }
    if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged)
        return fail(FEWBIT_ERR_INVALID_ARGUMENT, "%s (%p) is not device memory", what, p);
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, const_cast<void *>(p)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(FEWBIT_ERR_INVALID_ARGUMENT, "%s (%p): no allocation found", what, p);
    }
    const size_t off = static_cast<size_t>(static_cast<const char *>(p) - static_cast<const char *>(base));
    if (off + bytes > size)
        return fail(FEWBIT_ERR_INVALID_ARGUMENT, "%s needs %zu bytes but only %zu remain in its allocation", what, bytes, size - off);
    return FEWBIT_OK;
}

size_t dtype_size(int dtype) { return dtype == FEWBIT_F32 ? 4 : 2; }

#define FB_VALIDATE(WHAT, P, BYTES)                                  \
    do {                                                             \
        const int rc_ = validate_range((WHAT), (P), (BYTES));        \
        if (rc_ != FEWBIT_OK) return rc_;                            \
    } while (0)

long long tune(TuneKey k) {
    tune_init();
    return g_tune[k].load(std::memory_order_relaxed);
}

// ------------------------------------------------------------------------------------------------
// The device a call launches on = the calling thread's current device, resolved ONCE per call (the torch glue and the
// ctypes binding both make the tensors' device current first).  Geometry is cached per device index.
constexpr int kMaxDevices = 64;
struct Device { int index = 0, cus = 256; };

int get_device(Device &d) {
    static std::atomic<int> cus[kMaxDevices];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return fail(FEWBIT_ERR_LAUNCH, "hipGetDevice failed: no current HIP device");
    }
    if (dev < 0 || dev >= kMaxDevices) return fail(FEWBIT_ERR_UNSUPPORTED, "device index %d outside [0, %d)", dev, kMaxDevices);
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) {
            (void)hipGetLastError();
            v = 256;
        }
        cus[dev].store(v, std::memory_order_relaxed);     // (two threads racing on a first use store the same value)
    }
    d.index = dev;
    d.cus = v;
    return FEWBIT_OK;
}

// resident blocks per CU of a kernel instantiation: the runtime's occupancy answer, cached per kernel and device
template <auto Kern> int occupancy_blocks_per_cu(const Device &d, int threads) {
    static std::atomic<int> cached[kMaxDevices];
    int nb = cached[d.index].load(std::memory_order_relaxed);
    if (nb == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, Kern, threads, 0) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 1;
        }
        cached[d.index].store(nb, std::memory_order_relaxed);
    }
    return nb;
}

const char *const kFnNames[FEWBIT_CONTINUOUS_COUNT] = {"celu", "elu", "gelu", "hardswish", "logsigmoid", "mish", "selu", "sigmoid",
                                                       "silu", "softplus", "softsign", "tanh", "tanhshrink", "identity",
                                                       "identity_fold"};
const char *const kStepNames[FEWBIT_STEPWISE_COUNT] = {"hardshrink", "hardsigmoid", "hardtanh", "leaky_relu", "relu", "relu6",
                                                       "softshrink", "threshold"};
const char *dtype_name(int dt) { return dt == FEWBIT_F32 ? "f32" : dt == FEWBIT_F16 ? "f16" : "bf16"; }

// Launch shape of a streaming kernel (see Span): RESIDENT (chunk = 0: one wave per tile until the chip is full, then the
// resident waves loop round-robin) or CHUNKED (chunk = T tiles per wave, block-contiguous, as many blocks as that takes).
struct Shape { unsigned blocks; int chunk; };

// Built-in policy (measured on MI355X, scratch/headvar.py with SIZES=..., EXPERIMENTS.md section 6): the resident shape wins
// while a wave has only a few tiles (4096x4096 bf16: 11.2 vs 11.6 us), the chunked shape wins once the tensor is many
// times the resident generation, where statically assigned waves drift apart and the launch waits for the slowest
// (2^28 bf16 elements: backward 235 -> 197 us, pattern-table forward 224 -> 197 us; RoBERTa-size fp32, 5*10^7 elements:
// forward+backward 169 -> 151 us).  `min_ratio` = tiles per resident wave from which the chunked shape is used.
// `setting`: -1 = that policy, 0 = always resident, T = chunk of T wherever the tensor has more tiles than resident waves.
Shape launch_shape(size_t ntiles, int waves_per_block, size_t resident_blocks, long long setting, int auto_chunk, size_t min_ratio) {
    const size_t wpb = static_cast<size_t>(waves_per_block);
    size_t blocks = (ntiles + wpb - 1) / wpb;                      // one tile per wave
    int chunk = 0;
    if (blocks > resident_blocks) {
        long long t = setting;
        if (t < 0) t = ntiles >= min_ratio * resident_blocks * wpb ? auto_chunk : 0;
        const size_t chunked = t > 0 ? (ntiles + wpb * t - 1) / (wpb * t) : 0;
        if (t > 0 && chunked > resident_blocks) {
            blocks = chunked;
            chunk = static_cast<int>(t);
        } else {
            blocks = resident_blocks;
        }
    }
    if (blocks < 1) blocks = 1;
    return Shape{static_cast<unsigned>(blocks), chunk};
}

// search / backward / 1-bit kernels (cheap per-block setup): one tile per wave from 4 tiles per resident wave up;
// pattern-table forward (64 KiB table per block): three tiles per wave (an odd count: chunks of a power-of-two size start
// every block on the same memory channels, T = 4 measured 5-15 % slower than T = 3) from 12 tiles per resident wave up
constexpr int kAutoChunk = 1, kAutoLutChunk = 3;
constexpr size_t kAutoChunkRatio = 4, kAutoLutChunkRatio = 12;

// launch (or, dry, only describe) a 256-thread streaming kernel instantiation; the kernels' last parameter is the chunk
template <auto Kern, typename... Args>
void launch_tiled(Plan *plan, bool dry, const Device &dev, size_t n, int U, hipStream_t s, Args... args) {
    int per_cu = occupancy_blocks_per_cu<Kern>(dev, kBlock);
    if (per_cu > 8) per_cu = 8;
    const long long cap = tune(T_WAVES_PER_CU);
    if (cap >= kWavesPerBlock && cap / kWavesPerBlock < per_cu) per_cu = static_cast<int>(cap / kWavesPerBlock);
    const size_t ntiles = (n / 8) / (static_cast<size_t>(U) * kWave);
    const Shape sh = launch_shape(ntiles, kWavesPerBlock, static_cast<size_t>(dev.cus) * per_cu, tune(T_CHUNK), kAutoChunk,
                                  kAutoChunkRatio);
    if (plan) {
        plan->blocks = sh.blocks;
        plan->threads = kBlock;
        plan->chunk = sh.chunk;
        plan->u = U;
        plan->blocks_per_cu = per_cu;
    }
    if (!dry) hipLaunchKernelGGL(Kern, dim3(sh.blocks), dim3(kBlock), 0, s, args..., sh.chunk);
}

// pattern-table forward: BLOCK-thread blocks, at most two resident per CU (LDS), each wave loops over its tiles
template <auto Kern, int BLOCK, typename... Args>
void launch_lut(Plan *plan, bool dry, const Device &dev, size_t n, int U, hipStream_t s, Args... args) {
    int per_cu = occupancy_blocks_per_cu<Kern>(dev, BLOCK);
    if (per_cu > 2) per_cu = 2;
    const long long cap = tune(T_LUT_BLOCKS_PER_CU);
    if (cap >= 1 && cap < per_cu) per_cu = static_cast<int>(cap);
    const size_t ntiles = (n / 8) / (static_cast<size_t>(U) * kWave);
    const Shape sh = launch_shape(ntiles, BLOCK / kWave, static_cast<size_t>(dev.cus) * per_cu, tune(T_LUT_CHUNK),
                                  kAutoLutChunk, kAutoLutChunkRatio);
    if (plan) {
        plan->blocks = sh.blocks;
        plan->threads = BLOCK;
        plan->chunk = sh.chunk;
        plan->u = U;
        plan->blocks_per_cu = per_cu;
    }
    if (!dry) hipLaunchKernelGGL(Kern, dim3(sh.blocks), dim3(BLOCK), 0, s, args..., sh.chunk);
}

// Smallest tensor that takes the pattern-table forward: building the table costs the same whatever the table, the register
// search it replaces costs 2^k - 1 + k slow-class VALU per element -- measured crossover (scratch/xover.py, round 2):
// 6 Mi elements for k <= 3 (6.39 vs 6.43 us), 4.5 Mi for k = 4 (5 Mi: 6.48 vs 6.68 us).  lut_min overrides both; a huge
// value disables the kernel.
size_t lut_min_elements(int k) {
    const long long forced = tune(T_LUT_MIN);
    if (forced >= 0) return forced == 0 ? 1 : static_cast<size_t>(forced);
    return k == 4 ? (static_cast<size_t>(9) << 19) : (static_cast<size_t>(6) << 20);
}

// Groups per lane per pipeline stage, the values the policy uses: backward, 1-bit and fp32 forward 1 and 2; 16-bit forward 1
// (U = 4 never won a size class: profiles/r03_backward_shape_sweep.txt).
template <int... Us> struct UList {};
typedef UList<1> FwdUs16;
typedef UList<1, 2> FwdUs32;
typedef UList<1, 2> StreamUs16;
typedef UList<1, 2> StreamUs32;

// call f(integral_constant<U>) for the U of the list that `want` names (the list's first entry if it names none)
template <int U0, int... Us, typename F> void with_u(UList<U0, Us...>, long long want, F &&f) {
    bool done = false;
    auto one = [&](auto tag) {
        if (!done && want == decltype(tag)::value) {
            done = true;
            f(tag);
        }
    };
    one(std::integral_constant<int, U0>{});
    (one(std::integral_constant<int, Us>{}), ...);
    if (!done) f(std::integral_constant<int, U0>{});
}

// Built-in U policy (MI355X, round 3: profiles/r03_shape_sweep_*.txt, profiles/r03_backward_size_crossover.txt).
//   16-bit backward / 1-bit backward: two groups per lane per stage in the resident shape up to ~1.25x the headline size
//     (4096x4096: 11.04 us against 11.51 with one); from 20 Mi elements on ONE group per lane in the one-tile-per-wave
//     grid (chunk = 1 follows from the shape policy): cache-cold 3.5-4.6 % faster at every size measured (8192x4096 bf16
//     26.10 -> 24.91 us, 16384x3072 bf16 37.50 -> 35.99 us, 8192x8192 fp16 47.76 -> 45.73 us) for 0-2 % cache-warm --
//     inside a training step the tensors are cold (profiles/r03_roberta_kernel_stats_*.csv), so cold decides.
//   fp32 forward (register search): two groups per lane from 32 Mi elements on (16384x3072 fp32: 64.6 -> 61.0 us warm,
//     72.2 -> 71.0 us cold; at 4096x4096 one group is better when cold: 25.7 against 27.3 us).
//   everything else: one group per lane.
constexpr size_t kLargeStream16 = static_cast<size_t>(20) << 20;
constexpr size_t kLargeForward32 = static_cast<size_t>(32) << 20;
template <int DT> long long policy_u_fwd(size_t n) { return DT == FEWBIT_F32 && n >= kLargeForward32 ? 2 : 1; }
template <int DT> long long policy_u_bwd(size_t n) { return DT == FEWBIT_F32 || n >= kLargeStream16 ? 1 : 2; }
//   1-bit family (profiles/r03_shape_sweep_step1.txt, r03_shape_sweep_step1_sizes.txt; relu): these kernels are copies with a
//     compare, and two groups per lane is the better copy -- 16-bit forward from 6 Mi elements on (4096x4096 bf16: 11.02 ->
//     9.42 us warm, 13.65 -> 13.0 cold; 8192x4096: 20.79 -> 20.21 / 24.47 -> 23.64), 16-bit backward at every size
//     (25 Mi: 17.39 -> 16.29 / 20.29 -> 19.34); fp32 only between 4 Mi and 12 Mi elements (8 Mi: 9.74 -> 9.02 / 13.30 -> 12.81;
//     from 16 Mi on one group is better: 19.9 against 20.4 us).  Below ~4 Mi elements every shape takes the same 3.9 us.
template <int DT> bool step1_mid32(size_t n) { return n >= (static_cast<size_t>(4) << 20) && n < (static_cast<size_t>(12) << 20); }
template <int DT> long long policy_u_step1_fwd(size_t n) {
    if (DT == FEWBIT_F32) return step1_mid32<DT>(n) ? 2 : 1;
    return n >= (static_cast<size_t>(6) << 20) ? 2 : 1;
}
template <int DT> long long policy_u_step1_bwd(size_t n) { return DT == FEWBIT_F32 ? (step1_mid32<DT>(n) ? 2 : 1) : 2; }
long long tuned(TuneKey key, long long policy) {
    const long long t = tune(key);
    return t > 0 ? t : policy;
}

unsigned group_grid(size_t n) { return static_cast<unsigned>(((n + 7) / 8 + kBlock - 1) / kBlock); }

template <int FN, int DT>
int launch_forward(Plan *plan, bool dry, const void *x, void *y, uint8_t *state, size_t n, const void *borders, int nborders,
                   int k, float p0, float p1, hipStream_t s) {
    Device dev;
    if (const int rc = get_device(dev)) return rc;
    const bool pow2 = nborders == (1 << k) - 1;
    if (plan) plan->k = k;
    auto name = [&](const char *kern, int U, int block) {
        if (plan) snprintf(plan->kernel, sizeof plan->kernel, "%s<%s, %s, %d bits, U=%d, block=%d>", kern, kFnNames[FN], dtype_name(DT), k, U, block);
    };
    // (a folded key |x - shift| is an fp32 value, not one of the 65 536 input patterns: search kernels only)
    if constexpr (DT != FEWBIT_F32 && FN != FEWBIT_IDENTITY_FOLD) {
        // 16-bit dtypes, any table (power of two or not): pattern-table kernel once the tensor is big enough to pay for
        // building the table in every block
        if (n >= lut_min_elements(k)) {
            if (k > 4) {
                launch_lut<quantize_forward_lut_wide_kernel<FN, DT>, kLutBlock>(plan, dry, dev, n, 1, s, x, y, state, n, borders, nborders, k, p0, p1);
                name("quantize_forward_lut_wide_kernel", 1, kLutBlock);
                return dry ? FEWBIT_OK : check_launch("quantize_forward(lut)");
            }
            constexpr int U = 1, B = kLutBlock;      // one group per lane per stage, 1024-thread blocks (the only shape that ever won)
            switch (k) {
            case 1: launch_lut<quantize_forward_lut_kernel<FN, DT, 1, U, B>, B>(plan, dry, dev, n, U, s, x, y, state, n, borders, nborders, p0, p1); break;
            case 2: launch_lut<quantize_forward_lut_kernel<FN, DT, 2, U, B>, B>(plan, dry, dev, n, U, s, x, y, state, n, borders, nborders, p0, p1); break;
            case 3: launch_lut<quantize_forward_lut_kernel<FN, DT, 3, U, B>, B>(plan, dry, dev, n, U, s, x, y, state, n, borders, nborders, p0, p1); break;
            default: launch_lut<quantize_forward_lut_kernel<FN, DT, 4, U, B>, B>(plan, dry, dev, n, U, s, x, y, state, n, borders, nborders, p0, p1); break;
            }
            name("quantize_forward_lut_kernel", U, B);
            return dry ? FEWBIT_OK : check_launch("quantize_forward(lut)");
        }
    }
    if (pow2 && k <= 4) {          // full 1..4-bit tables: borders in registers
        typedef typename std::conditional<DT == FEWBIT_F32, FwdUs32, FwdUs16>::type Us;
        with_u(Us{}, tuned(T_U_FWD, policy_u_fwd<DT>(n)), [&](auto tag) {
            constexpr int U = decltype(tag)::value;
            switch (k) {
            case 1: launch_tiled<quantize_forward_kernel<FN, DT, 1, U>>(plan, dry, dev, n, U, s, x, y, state, n, borders, p0, p1); break;
            case 2: launch_tiled<quantize_forward_kernel<FN, DT, 2, U>>(plan, dry, dev, n, U, s, x, y, state, n, borders, p0, p1); break;
            case 3: launch_tiled<quantize_forward_kernel<FN, DT, 3, U>>(plan, dry, dev, n, U, s, x, y, state, n, borders, p0, p1); break;
            default: launch_tiled<quantize_forward_kernel<FN, DT, 4, U>>(plan, dry, dev, n, U, s, x, y, state, n, borders, p0, p1); break;
            }
            name("quantize_forward_kernel", U, kBlock);
        });
    } else {                       // 5..8-bit tables and tables that do not fill their bit width: borders in LDS
        launch_tiled<quantize_forward_wide_kernel<FN, DT>>(plan, dry, dev, n, 1, s, x, y, state, n, borders, nborders, k, p0, p1);
        name("quantize_forward_wide_kernel", 1, kBlock);
    }
    return dry ? FEWBIT_OK : check_launch("quantize_forward");
}

}  // namespace

// one functor, every dtype: the unit of the split build (external, hidden linkage; see FEWBIT_TU above)
template <int FN>
FEWBIT_HIDDEN int dispatch_forward_dtype(Plan *plan, bool dry, int dtype, const void *x, void *y, uint8_t *state, size_t n,
                                         const void *borders, int nborders, int k, float p0, float p1, hipStream_t s) {
    switch (dtype) {
    case FEWBIT_F32: return launch_forward<FN, FEWBIT_F32>(plan, dry, x, y, state, n, borders, nborders, k, p0, p1, s);
    case FEWBIT_F16: return launch_forward<FN, FEWBIT_F16>(plan, dry, x, y, state, n, borders, nborders, k, p0, p1, s);
    case FEWBIT_BF16: return launch_forward<FN, FEWBIT_BF16>(plan, dry, x, y, state, n, borders, nborders, k, p0, p1, s);
    default: return fail(FEWBIT_ERR_INVALID_ARGUMENT, "unknown dtype %d", dtype);
    }
}

#ifdef FEWBIT_TU
// functor id F is compiled by unit 1 + F / 3; every other unit only declares it
#define FB_DISPATCH_ARGS (Plan *, bool, int, const void *, void *, uint8_t *, size_t, const void *, int, int, float, float, hipStream_t)
#define FB_DECLARE(F) extern template int dispatch_forward_dtype<F> FB_DISPATCH_ARGS;
#define FB_DEFINE(F) template int dispatch_forward_dtype<F> FB_DISPATCH_ARGS;
#if FEWBIT_TU == 1
FB_DEFINE(0) FB_DEFINE(1) FB_DEFINE(2)
#else
FB_DECLARE(0) FB_DECLARE(1) FB_DECLARE(2)
#endif
#if FEWBIT_TU == 2
FB_DEFINE(3) FB_DEFINE(4) FB_DEFINE(5)
#else
FB_DECLARE(3) FB_DECLARE(4) FB_DECLARE(5)
#endif
#if FEWBIT_TU == 3
FB_DEFINE(6) FB_DEFINE(7) FB_DEFINE(8)
#else
FB_DECLARE(6) FB_DECLARE(7) FB_DECLARE(8)
#endif
#if FEWBIT_TU == 4
FB_DEFINE(9) FB_DEFINE(10) FB_DEFINE(11)
#else
FB_DECLARE(9) FB_DECLARE(10) FB_DECLARE(11)
#endif
#if FEWBIT_TU == 5
FB_DEFINE(12) FB_DEFINE(13) FB_DEFINE(14)
#else
FB_DECLARE(12) FB_DECLARE(13) FB_DECLARE(14)
#endif
static_assert(FEWBIT_CONTINUOUS_COUNT == 3 * FEWBIT_TU_COUNT, "every continuous functor needs a unit");
#endif

#if FEWBIT_CORE_TU
namespace {

template <int DT>
int launch_backward(Plan *plan, bool dry, const void *gy, const uint8_t *state, void *gx, size_t n, const void *levels, int nlevels,
                    int k, hipStream_t s) {
    Device dev;
    if (const int rc = get_device(dev)) return rc;
    if (plan) plan->k = k;
    if (k > 4) {
        launch_tiled<quantize_backward_wide_kernel<DT>>(plan, dry, dev, n, 1, s, gy, state, gx, n, levels, nlevels, k);
        if (plan) snprintf(plan->kernel, sizeof plan->kernel, "quantize_backward_wide_kernel<%s, %d bits>", dtype_name(DT), k);
        return dry ? FEWBIT_OK : check_launch("quantize_backward");
    }
    typedef typename std::conditional<DT == FEWBIT_F32, StreamUs32, StreamUs16>::type Us;
    with_u(Us{}, tuned(T_U_BWD, policy_u_bwd<DT>(n)), [&](auto tag) {
        constexpr int U = decltype(tag)::value;
        switch (k) {
        case 1: launch_tiled<quantize_backward_kernel<DT, 1, U>>(plan, dry, dev, n, U, s, gy, state, gx, n, levels, nlevels); break;
        case 2: launch_tiled<quantize_backward_kernel<DT, 2, U>>(plan, dry, dev, n, U, s, gy, state, gx, n, levels, nlevels); break;
        case 3: launch_tiled<quantize_backward_kernel<DT, 3, U>>(plan, dry, dev, n, U, s, gy, state, gx, n, levels, nlevels); break;
        default: launch_tiled<quantize_backward_kernel<DT, 4, U>>(plan, dry, dev, n, U, s, gy, state, gx, n, levels, nlevels); break;
        }
        if (plan) snprintf(plan->kernel, sizeof plan->kernel, "quantize_backward_kernel<%s, %d bits, U=%d>", dtype_name(DT), k, U);
    });
    return dry ? FEWBIT_OK : check_launch("quantize_backward");
}

template <int FN, int DT>
int launch_step1_forward(Plan *plan, bool dry, const void *x, void *y, uint8_t *state, size_t n, float p0, float p1, hipStream_t s) {
    Device dev;
    if (const int rc = get_device(dev)) return rc;
    if (plan) plan->k = 1;
    typedef typename std::conditional<DT == FEWBIT_F32, StreamUs32, StreamUs16>::type Us;
    with_u(Us{}, tuned(T_U_STEP1, policy_u_step1_fwd<DT>(n)), [&](auto tag) {
        constexpr int U = decltype(tag)::value;
        launch_tiled<stepwise1_forward_kernel<FN, DT, U>>(plan, dry, dev, n, U, s, x, y, state, n, p0, p1);
        if (plan) snprintf(plan->kernel, sizeof plan->kernel, "stepwise1_forward_kernel<%s, %s, U=%d>", kStepNames[FN], dtype_name(DT), U);
    });
    return dry ? FEWBIT_OK : check_launch("stepwise1_forward");
}

template <int FN>
int dispatch_step1_dtype(Plan *plan, bool dry, int dtype, const void *x, void *y, uint8_t *state, size_t n, float p0, float p1,
                         hipStream_t s) {
    switch (dtype) {
    case FEWBIT_F32: return launch_step1_forward<FN, FEWBIT_F32>(plan, dry, x, y, state, n, p0, p1, s);
    case FEWBIT_F16: return launch_step1_forward<FN, FEWBIT_F16>(plan, dry, x, y, state, n, p0, p1, s);
    case FEWBIT_BF16: return launch_step1_forward<FN, FEWBIT_BF16>(plan, dry, x, y, state, n, p0, p1, s);
    default: return fail(FEWBIT_ERR_INVALID_ARGUMENT, "unknown dtype %d", dtype);
    }
}

template <int DT>
int launch_step1_backward(Plan *plan, bool dry, const void *gy, const uint8_t *state, void *gx, size_t n, float m0, float m1,
                          hipStream_t s) {
    Device dev;
    if (const int rc = get_device(dev)) return rc;
    if (plan) plan->k = 1;
    typedef typename std::conditional<DT == FEWBIT_F32, StreamUs32, StreamUs16>::type Us;
    with_u(Us{}, tuned(T_U_STEP1, policy_u_step1_bwd<DT>(n)), [&](auto tag) {
        constexpr int U = decltype(tag)::value;
        launch_tiled<stepwise1_backward_kernel<DT, U>>(plan, dry, dev, n, U, s, gy, state, gx, n, m0, m1);
        if (plan) snprintf(plan->kernel, sizeof plan->kernel, "stepwise1_backward_kernel<%s, U=%d>", dtype_name(DT), U);
    });
    return dry ? FEWBIT_OK : check_launch("stepwise1_backward");
}

#define FB_CONTINUOUS_CASES                                                                                  \
    FB_CASE(FEWBIT_CELU) FB_CASE(FEWBIT_ELU) FB_CASE(FEWBIT_GELU) FB_CASE(FEWBIT_HARDSWISH)                 \
    FB_CASE(FEWBIT_LOGSIGMOID) FB_CASE(FEWBIT_MISH) FB_CASE(FEWBIT_SELU) FB_CASE(FEWBIT_SIGMOID)            \
    FB_CASE(FEWBIT_SILU) FB_CASE(FEWBIT_SOFTPLUS) FB_CASE(FEWBIT_SOFTSIGN) FB_CASE(FEWBIT_TANH)             \
    FB_CASE(FEWBIT_TANHSHRINK) FB_CASE(FEWBIT_IDENTITY) FB_CASE(FEWBIT_IDENTITY_FOLD)
#define FB_STEPWISE_CASES                                                                                    \
    FB_CASE(FEWBIT_HARDSHRINK) FB_CASE(FEWBIT_HARDSIGMOID) FB_CASE(FEWBIT_HARDTANH) FB_CASE(FEWBIT_LEAKY_RELU) \
    FB_CASE(FEWBIT_RELU) FB_CASE(FEWBIT_RELU6) FB_CASE(FEWBIT_SOFTSHRINK) FB_CASE(FEWBIT_THRESHOLD)

// ---- the four entry points with a `dry` flag (describe = the same dispatch without the launch) ----
int do_quantize_forward(Plan *plan, bool dry, int fn, int dtype, const void *x, void *y, uint8_t *state, size_t n,
                        const void *borders, int nborders, double p0, double p1, void *stream) {
    if (nborders < 1 || nborders > 255) return fail(FEWBIT_ERR_UNSUPPORTED, "nborders=%d outside [1,255]", nborders);
    if (dtype != FEWBIT_F32 && dtype != FEWBIT_F16 && dtype != FEWBIT_BF16) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "unknown dtype %d", dtype);
    const int k = fewbit_hip_bitwidth(nborders + 1);
    if (!dry) {
        if (n == 0) return FEWBIT_OK;
        if (!x || !y || !state || !borders) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "null pointer argument");
        FB_VALIDATE("x", x, n * dtype_size(dtype));
        FB_VALIDATE("y", y, n * dtype_size(dtype));
        FB_VALIDATE("state", state, fewbit_hip_state_nbytes(n, k));
        FB_VALIDATE("borders", borders, static_cast<size_t>(nborders) * dtype_size(dtype));
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float a = static_cast<float>(p0), b = static_cast<float>(p1);
#define FB_CASE(F) case F: return dispatch_forward_dtype<F>(plan, dry, dtype, x, y, state, n, borders, nborders, k, a, b, s);
    switch (fn) {
        FB_CONTINUOUS_CASES
    default: return fail(FEWBIT_ERR_INVALID_ARGUMENT, "unknown continuous fn %d", fn);
    }
#undef FB_CASE
}

int do_quantize_backward(Plan *plan, bool dry, int dtype, const void *gy, const uint8_t *state, void *gx, size_t n,
                         const void *levels, int nlevels, void *stream) {
    if (nlevels < 2 || nlevels > 256) return fail(FEWBIT_ERR_UNSUPPORTED, "nlevels=%d outside [2,256]", nlevels);
    if (dtype != FEWBIT_F32 && dtype != FEWBIT_F16 && dtype != FEWBIT_BF16) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "unknown dtype %d", dtype);
    const int k = fewbit_hip_bitwidth(nlevels);
    if (!dry) {
        if (n == 0) return FEWBIT_OK;
        if (!gy || !gx || !state || !levels) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "null pointer argument");
        FB_VALIDATE("gy", gy, n * dtype_size(dtype));
        FB_VALIDATE("gx", gx, n * dtype_size(dtype));
        FB_VALIDATE("state", state, fewbit_hip_state_nbytes(n, k));
        FB_VALIDATE("levels", levels, static_cast<size_t>(nlevels) * dtype_size(dtype));
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case FEWBIT_F32: return launch_backward<FEWBIT_F32>(plan, dry, gy, state, gx, n, levels, nlevels, k, s);
    case FEWBIT_F16: return launch_backward<FEWBIT_F16>(plan, dry, gy, state, gx, n, levels, nlevels, k, s);
    default: return launch_backward<FEWBIT_BF16>(plan, dry, gy, state, gx, n, levels, nlevels, k, s);
    }
}

int do_stepwise1_forward(Plan *plan, bool dry, int fn, int dtype, const void *x, void *y, uint8_t *state, size_t n, double p0,
                         double p1, void *stream) {
    if (dtype != FEWBIT_F32 && dtype != FEWBIT_F16 && dtype != FEWBIT_BF16) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "unknown dtype %d", dtype);
    if (!dry) {
        if (n == 0) return FEWBIT_OK;
        if (!x || !y || !state) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "null pointer argument");
        FB_VALIDATE("x", x, n * dtype_size(dtype));
        FB_VALIDATE("y", y, n * dtype_size(dtype));
        FB_VALIDATE("state", state, fewbit_hip_state_nbytes(n, 1));
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float a = static_cast<float>(p0), b = static_cast<float>(p1);
#define FB_CASE(F) case F: return dispatch_step1_dtype<F>(plan, dry, dtype, x, y, state, n, a, b, s);
    switch (fn) {
        FB_STEPWISE_CASES
    default: return fail(FEWBIT_ERR_INVALID_ARGUMENT, "unknown stepwise fn %d", fn);
    }
#undef FB_CASE
}

int do_stepwise1_backward(Plan *plan, bool dry, int fn, int dtype, const void *gy, const uint8_t *state, void *gx, size_t n,
                          double p0, void *stream) {
    if (fn < 0 || fn >= FEWBIT_STEPWISE_COUNT) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "unknown stepwise fn %d", fn);
    if (dtype != FEWBIT_F32 && dtype != FEWBIT_F16 && dtype != FEWBIT_BF16) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "unknown dtype %d", dtype);
    if (!dry) {
        if (n == 0) return FEWBIT_OK;
        if (!gy || !gx || !state) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "null pointer argument");
        FB_VALIDATE("gy", gy, n * dtype_size(dtype));
        FB_VALIDATE("gx", gx, n * dtype_size(dtype));
        FB_VALIDATE("state", state, fewbit_hip_state_nbytes(n, 1));
    }
    float m0 = 0.0f, m1 = 1.0f;
    if (fn == FEWBIT_HARDSIGMOID) m1 = 1.0f / 6.0f;
    if (fn == FEWBIT_LEAKY_RELU) { m0 = 1.0f; m1 = static_cast<float>(p0); }
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
    case FEWBIT_F32: return launch_step1_backward<FEWBIT_F32>(plan, dry, gy, state, gx, n, m0, m1, s);
    case FEWBIT_F16: return launch_step1_backward<FEWBIT_F16>(plan, dry, gy, state, gx, n, m0, m1, s);
    default: return launch_step1_backward<FEWBIT_BF16>(plan, dry, gy, state, gx, n, m0, m1, s);
    }
}

int write_plan(const Plan &p, char *buf, size_t len) {
    if (!buf || len == 0) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "describe: no output buffer");
    snprintf(buf, len, "{\"kernel\": \"%s\", \"blocks\": %u, \"threads\": %d, \"blocks_per_cu\": %d, \"chunk\": %d, \"u\": %d, \"bits\": %d}",
             p.kernel, p.blocks, p.threads, p.blocks_per_cu, p.chunk, p.u, p.k);
    return FEWBIT_OK;
}

}  // namespace
#endif  // FEWBIT_CORE_TU
}  // namespace fewbit_hip

#if FEWBIT_CORE_TU
// ================================================================================================
// C-ABI
// ================================================================================================
using namespace fewbit_hip;

extern "C" {

int fewbit_hip_abi_version(void) { return FEWBIT_HIP_ABI_VERSION; }

const char *fewbit_hip_last_error(void) { return g_last_error; }

int fewbit_hip_bitwidth(int nlevels) {
    int k = 0;
    while ((1 << k) < nlevels) ++k;
    return k < 1 ? 1 : k;
}

size_t fewbit_hip_state_nbytes(size_t n, int nbits) { return static_cast<size_t>(nbits) * ((n + 7) / 8); }

int fewbit_hip_quantize_forward(int fn, int dtype, const void *x, void *y, uint8_t *state, size_t n,
                                const void *borders, int nborders, double p0, double p1, void *stream) {
    return do_quantize_forward(nullptr, false, fn, dtype, x, y, state, n, borders, nborders, p0, p1, stream);
}

int fewbit_hip_quantize_backward(int dtype, const void *gy, const uint8_t *state, void *gx, size_t n,
                                 const void *levels, int nlevels, void *stream) {
    return do_quantize_backward(nullptr, false, dtype, gy, state, gx, n, levels, nlevels, stream);
}

int fewbit_hip_stepwise1_forward(int fn, int dtype, const void *x, void *y, uint8_t *state, size_t n, double p0,
                                 double p1, void *stream) {
    return do_stepwise1_forward(nullptr, false, fn, dtype, x, y, state, n, p0, p1, stream);
}

int fewbit_hip_stepwise1_backward(int fn, int dtype, const void *gy, const uint8_t *state, void *gx, size_t n,
                                  double p0, void *stream) {
    return do_stepwise1_backward(nullptr, false, fn, dtype, gy, state, gx, n, p0, stream);
}

int fewbit_hip_describe_quantize_forward(int fn, int dtype, size_t n, int nborders, char *buf, size_t len) {
    Plan p{};
    const int rc = do_quantize_forward(&p, true, fn, dtype, nullptr, nullptr, nullptr, n, nullptr, nborders, 0.0, 0.0, nullptr);
    return rc ? rc : write_plan(p, buf, len);
}

int fewbit_hip_describe_quantize_backward(int dtype, size_t n, int nlevels, char *buf, size_t len) {
    Plan p{};
    const int rc = do_quantize_backward(&p, true, dtype, nullptr, nullptr, nullptr, n, nullptr, nlevels, nullptr);
    return rc ? rc : write_plan(p, buf, len);
}

int fewbit_hip_describe_stepwise1_forward(int fn, int dtype, size_t n, char *buf, size_t len) {
    Plan p{};
    const int rc = do_stepwise1_forward(&p, true, fn, dtype, nullptr, nullptr, nullptr, n, 0.0, 0.0, nullptr);
    return rc ? rc : write_plan(p, buf, len);
}

int fewbit_hip_describe_stepwise1_backward(int fn, int dtype, size_t n, char *buf, size_t len) {
    Plan p{};
    const int rc = do_stepwise1_backward(&p, true, fn, dtype, nullptr, nullptr, nullptr, n, 0.0, nullptr);
    return rc ? rc : write_plan(p, buf, len);
}

int fewbit_hip_tune(const char *key, long long value) {
    if (!key) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "tune: null key");
    tune_init();
    for (int i = 0; i < T_COUNT; ++i) {
        if (strcmp(key, kTuneSpec[i].key) == 0) {
            g_tune[i].store(value, std::memory_order_relaxed);
            return FEWBIT_OK;
        }
    }
    bool known = false;                    // (the random-projection unit's own keys: fewbit_sketch.hip)
    const int rc = sketch_tune(key, value, &known);
    if (known) return rc;
    return fail(FEWBIT_ERR_INVALID_ARGUMENT, "tune: unknown key '%s'", key);
}

int fewbit_hip_pack_codes(const int32_t *codes, uint8_t *state, size_t n, int nbits, void *stream) {
    if (nbits < 1 || nbits > 8) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "nbits=%d outside [1,8]", nbits);
    if (n == 0) return FEWBIT_OK;
    if (!codes || !state) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "null pointer argument");
    FB_VALIDATE("codes", codes, n * sizeof(int32_t));
    FB_VALIDATE("state", state, fewbit_hip_state_nbytes(n, nbits));
    hipLaunchKernelGGL(pack_codes_kernel, dim3(group_grid(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), codes,
                       state, n, nbits);
    return check_launch("pack_codes");
}

int fewbit_hip_unpack_codes(const uint8_t *state, int32_t *codes, size_t n, int nbits, void *stream) {
    if (nbits < 1 || nbits > 8) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "nbits=%d outside [1,8]", nbits);
    if (n == 0) return FEWBIT_OK;
    if (!codes || !state) return fail(FEWBIT_ERR_INVALID_ARGUMENT, "null pointer argument");
    FB_VALIDATE("codes", codes, n * sizeof(int32_t));
    FB_VALIDATE("state", state, fewbit_hip_state_nbytes(n, nbits));
    hipLaunchKernelGGL(unpack_codes_kernel, dim3(group_grid(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       state, codes, n, nbits);
    return check_launch("unpack_codes");
}

}  // extern "C"
#endif  // FEWBIT_CORE_TU
