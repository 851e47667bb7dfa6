/*
 * fewbit_hip.h -- C-ABI of the MI355X (gfx950) implementation of FewBit's
 * quantized-activation path.  Plain pointers and sizes, no torch types.
 *
 * Every pointer is a DEVICE pointer; `stream` is a hipStream_t passed as
 * void* (NULL = the null stream).  Calls only enqueue work on `stream`; they
 * never synchronise.  Return value: FEWBIT_OK or a negative fewbit_status;
 * fewbit_hip_last_error() gives the message of the last failure on the
 * calling thread.
 *
 * Which reference interface each entry point replaces (paths relative to the
 * reference tree skolai/fewbit):
 *
 *   fewbit_hip_quantize_forward   <- the 13 `DECLARE_CONTINOUS_FUNC` launchers
 *                                    Celu ... Tanhshrink, fewbit/cuda/codec.h:75-92
 *                                    (kernel StepwiseKernel, fewbit/cuda/codec.cu:489-504)
 *   fewbit_hip_quantize_backward  <- StepwiseBackward, fewbit/cuda/codec.h:94-96
 *                                    (fewbit/cuda/codec.cu:655-670)
 *   fewbit_hip_stepwise1_forward  <- Hardshrink/Hardsigmoid/Hardtanh/LeakyRelu/Relu/Relu6/
 *                                    Softshrink/Threshold, fewbit/cuda/codec.h:59-68
 *   fewbit_hip_stepwise1_backward <- <Name>Backward, fewbit/cuda/codec.h:59-68
 *   fewbit_hip_pack_codes         <- DeflateBlock, fewbit/cuda/codec.h:16-17 (layout of
 *                                    fewbit::Deflate, fewbit/cpu/codec.h:33-57)
 *   fewbit_hip_unpack_codes       <- InflateBlock, fewbit/cuda/codec.h:21-22
 *   fewbit_hip_state_nbytes       <- buffer_len = nobits * ceil(n/8),
 *                                    fewbit/cuda/activation.cc:349-351
 *   fewbit_hip_bitwidth           <- GetBitWidth, fewbit/cuda/activation.cc:17-21 (the
 *                                    off-by-one of that function is NOT reproduced; this is
 *                                    ceil(log2(nlevels)) as on the CPU path, fewbit/cpu/gelu.cc:18,36)
 *
 * Differences from the reference launchers, all additive: fp16/bf16 I/O besides
 * fp32, 64-bit element counts, an explicit stream, and error returns.
 *
 * Pointers are trusted by default (no runtime query on the launch path).  With
 * FEWBIT_HIP_VALIDATE=1 in the environment every pointer is checked to be
 * device memory and every buffer to extend far enough inside its allocation
 * (the underlying hipMalloc: sub-blocks of a caching allocator share one);
 * violations return FEWBIT_ERR_INVALID_ARGUMENT.
 *
 * Packed state layout (identical to the reference): element i occupies bits
 * [k*i, k*(i+1)) of one little-endian, LSB-first bitstream, so every 8
 * consecutive elements map to exactly k bytes.  The state buffer is
 * k*ceil(n/8) bytes; codes of the padding elements are zero.
 */
#ifndef FEWBIT_HIP_H_
#define FEWBIT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version history.  Bindings check it and refuse an older library by name.
 *   1: the quantized-activation path.
 *   2: + fewbit_hip_describe_*, fewbit_hip_tune and the random-projection entry points fewbit_hip_sketch*.
 *   3: + seeds in device memory (fewbit_hip_sketch_device_seed, fewbit_hip_sketch_next_seed, fewbit_hip_sketch_mix_seed).
 *   4: + fewbit_hip_xoshiro128pp; the Gaussian S redefined on xoshiro128++ streams (the same seed gives another matrix than under 3).
 *   5: FROZEN.  The six fewbit_hip_sketch_tune_* measurement hooks of versions 2-4 are gone from the interface: their settings are
 *      keys of the one remaining hook, fewbit_hip_tune ("sketch_slices", ...); + fewbit_hip_sampled_dct, _seeded, fewbit_hip_sampled_rows
 *      (the reference's 'dct' estimator).  What is declared below is what a binding needs (tests/test_api.py pins the exported symbol list). */
#define FEWBIT_HIP_ABI_VERSION 5

typedef enum fewbit_status {
    FEWBIT_OK = 0,
    FEWBIT_ERR_INVALID_ARGUMENT = -1, /* bad enum, null pointer, table size, misuse */
    FEWBIT_ERR_UNSUPPORTED = -2,      /* e.g. more than 256 levels */
    FEWBIT_ERR_LAUNCH = -3            /* HIP reported an error at launch */
} fewbit_status;

typedef enum fewbit_dtype { FEWBIT_F32 = 0, FEWBIT_F16 = 1, FEWBIT_BF16 = 2 } fewbit_dtype;

/* continuous activations: k-bit code from a border table (fewbit/fewbit.cc:21-33) */
typedef enum fewbit_continuous_fn {
    FEWBIT_CELU = 0,     /* p0 = alpha */
    FEWBIT_ELU = 1,      /* p0 = alpha */
    FEWBIT_GELU = 2,
    FEWBIT_HARDSWISH = 3,
    FEWBIT_LOGSIGMOID = 4,
    FEWBIT_MISH = 5,
    FEWBIT_SELU = 6,
    FEWBIT_SIGMOID = 7,
    FEWBIT_SILU = 8,
    FEWBIT_SOFTPLUS = 9, /* p0 = beta, p1 = threshold */
    FEWBIT_SOFTSIGN = 10,
    FEWBIT_TANH = 11,
    FEWBIT_TANHSHRINK = 12,
    FEWBIT_IDENTITY = 13, /* y = x: quantize only (custom `stepwise` tables) */
    /* y = x, code from the FOLDED key |x - p0| (fp32): the even-`parity` form of the custom `stepwise` table that the
     * reference declares (fewbit/fewbit.cc:37, fewbit/modules/activations.py:97-134) and never implements; the
     * borders are those of the half line t >= 0 (fewbit/approx.py:92-101, domain (0, x_max)), p0 = shift_x */
    FEWBIT_IDENTITY_FOLD = 14,
    FEWBIT_CONTINUOUS_COUNT = 15
} fewbit_continuous_fn;

/* piecewise-linear activations: exact 1-bit state (fewbit/fewbit.cc:10-18) */
typedef enum fewbit_stepwise_fn {
    FEWBIT_HARDSHRINK = 0, /* p0 = lambd */
    FEWBIT_HARDSIGMOID = 1,
    FEWBIT_HARDTANH = 2,   /* p0 = min_val, p1 = max_val */
    FEWBIT_LEAKY_RELU = 3, /* p0 = negative_slope (forward AND backward) */
    FEWBIT_RELU = 4,
    FEWBIT_RELU6 = 5,
    FEWBIT_SOFTSHRINK = 6, /* p0 = lambd */
    FEWBIT_THRESHOLD = 7,  /* p0 = threshold, p1 = value */
    FEWBIT_STEPWISE_COUNT = 8
} fewbit_stepwise_fn;

int fewbit_hip_abi_version(void);
const char *fewbit_hip_last_error(void);

/* k = ceil(log2(nlevels)), at least 1 */
int fewbit_hip_bitwidth(int nlevels);
/* k * ceil(n/8) */
size_t fewbit_hip_state_nbytes(size_t n, int nbits);

/*
 * Fused forward: y[i] = fn(x[i]); code[i] = #{ j : borders[j] < x[i] } (NaN -> nborders);
 * state = packed k-bit codes, k = fewbit_hip_bitwidth(nborders + 1).
 *   x, y     n elements of `dtype`; y may alias x (in-place, as the reference op does)
 *   state    fewbit_hip_state_nbytes(n, k) bytes, fully overwritten
 *   borders  nborders INNER borders, ascending, same dtype as x (1 <= nborders <= 255)
 */
int fewbit_hip_quantize_forward(int fn, int dtype, const void *x, void *y, uint8_t *state, size_t n,
                                const void *borders, int nborders, double p0, double p1, void *stream);

/*
 * Fused backward: gx[i] = levels[code[i]] * gy[i], product in fp32, rounded to nearest even.
 *   levels   nlevels values of `dtype` (2 <= nlevels <= 256); k = fewbit_hip_bitwidth(nlevels)
 *   gx may alias gy.
 */
int fewbit_hip_quantize_backward(int dtype, const void *gy, const uint8_t *state, void *gx, size_t n,
                                 const void *levels, int nlevels, void *stream);

/* 1-bit family forward: y = fn(x), state bit i = "derivative is the non-default branch" */
int fewbit_hip_stepwise1_forward(int fn, int dtype, const void *x, void *y, uint8_t *state, size_t n, double p0,
                                 double p1, void *stream);

/* 1-bit family backward: gx = (bit ? m1 : m0) * gy; (m0,m1) = (0,1), hardsigmoid (0,1/6), leaky_relu (1,p0) */
int fewbit_hip_stepwise1_backward(int fn, int dtype, const void *gy, const uint8_t *state, void *gx, size_t n,
                                  double p0, void *stream);

/*
 * What a call WOULD launch, without launching it: the same dispatch as the entry point of the same name, for `n`
 * elements on the calling thread's current device, written to `buf` as one JSON object
 *   {"kernel": "...", "blocks": B, "threads": T, "blocks_per_cu": C, "chunk": X, "u": U, "bits": k}
 * (bench.py takes the name of the dominant kernel from here; the reference has no counterpart -- its launch
 * topology is one macro, fewbit/cuda/codec.cu:229-239).
 */
int fewbit_hip_describe_quantize_forward(int fn, int dtype, size_t n, int nborders, char *buf, size_t len);
int fewbit_hip_describe_quantize_backward(int dtype, size_t n, int nlevels, char *buf, size_t len);
int fewbit_hip_describe_stepwise1_forward(int fn, int dtype, size_t n, char *buf, size_t len);
int fewbit_hip_describe_stepwise1_backward(int fn, int dtype, size_t n, char *buf, size_t len);

/*
 * The one tuning hook (tests and measurement scripts; never needed for correctness): value -1 restores the built-in policy.
 *   activation kernels -- every setting computes the same bytes: "waves_per_cu", "chunk", "lut_chunk", "lut_blocks_per_cu",
 *     "lut_min", "u_fwd", "u_bwd", "u_step1" (also read once from the environment, FEWBIT_HIP_<KEY in upper case>, at the first launch);
 *   random-projection kernels -- every setting computes the same sums up to the fp32 association across row slices and the
 *     rounding of bf16 partial sums: "sketch_slices" (> 0), "sketch_waves" (4 | 8), "sketch_halves" (1 | 2), "sketch_convert"
 *     (0 | 1: fp32 input rounded to bf16 in one pass first), "sketch_partials" (0 | 1 | 2: bf16 partial sums never / policy / for
 *     bf16 results only), "sketch_materialise" (0 | 1: Gaussian S never through memory / also on narrow fp32 layers);
 *     fewbit_hip_sketch_workspace and fewbit_hip_sketch_describe follow the settings.
 * Unknown keys and values outside a key's set return FEWBIT_ERR_INVALID_ARGUMENT.
 */
int fewbit_hip_tune(const char *key, long long value);

/* stand-alone codec (test seam): int32 codes <-> packed state, 1 <= nbits <= 8 */
int fewbit_hip_pack_codes(const int32_t *codes, uint8_t *state, size_t n, int nbits, void *stream);
int fewbit_hip_unpack_codes(const uint8_t *state, int32_t *codes, size_t n, int nbits, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Random-projection products of the randomized linear layers:  out = scale * S . M
 *   replaces  `proj = T.randn((proj_features, rows)); proj @ input_view`  and the Rademacher twin,
 *             fewbit/functional/linear.py:133-146 (forward) and :195-208 (backward, which re-draws the same matrix
 *             from the saved generator state)
 *   S    proj x rows, a pure function of (seed, row, column) in the operand layout of the matrix cores (definition:
 *        fewbit_amd/csrc/fewbit_sketch.hip header; host model: tests/sketch_reference.py).  Rademacher: evaluated in registers inside
 *        the product kernel, never in memory.  Gaussian, operand wider than 256 features: written ONCE PER CALL into the workspace as
 *        MFMA fragments and read back by the product kernel (fewbit_hip_sketch_describe reports the bytes as "s_fragment_bytes";
 *        0 = generated in registers); never kept beyond the call.  Forward and backward pass the same seed and get the same S.
 *   m    rows x features, row-major with leading dimension `ld` (elements), dtype F32 / F16 / BF16; fp32 is rounded to bf16
 *        (while it is staged, or for many row tiles in one pass beforehand) -- BF16 OPERANDS: the products run on the bf16 matrix
 *        pipe and are accumulated in fp32 (the reference multiplies fp32 by fp32); when the rows are sliced and the operands are
 *        bf16, each slice's sum crosses the workspace rounded to bf16 and the slices are added in fp32 (tune key "sketch_partials")
 *   out  proj x features, contiguous, the dtype of m
 *   workspace  fewbit_hip_sketch_workspace(dist, dtype, rows, features, proj) bytes of device memory (0 for small unsliced calls, a few
 *        hundred MB for a large Gaussian one); contents are scratch.  The result is deterministic: the same arguments give the same bits.
 */
typedef enum fewbit_sketch_dist { FEWBIT_SKETCH_RADEMACHER = 0, FEWBIT_SKETCH_GAUSSIAN = 1 } fewbit_sketch_dist;

size_t fewbit_hip_sketch_workspace(int dist, int dtype, size_t rows, size_t features, size_t proj);
int fewbit_hip_sketch(int dist, int dtype, const void *m, size_t rows, size_t features, size_t ld, size_t proj, uint64_t seed,
                      double scale, void *out, void *workspace, size_t workspace_bytes, void *stream);
/* The same product with the seed read from DEVICE memory when the kernel runs.  A launch recorded in a hipGraph replays its
 * arguments, so a seed passed by value would give every replay the same S (the reference cannot be captured at all: it reads
 * the generator state back, fewbit/functional/linear.py:105); here the recorded pair
 *     fewbit_hip_sketch_next_seed(counter, base, seed)      *seed = fewbit_hip_sketch_mix_seed(base, (*counter)++)
 *     fewbit_hip_sketch_device_seed(..., seed, ...)
 * draws a fresh matrix on every replay, and the layer's backward re-reads the `seed` word its forward left behind.
 * `counter`, `seed_device`: 8-byte aligned device words owned by the caller. */
int fewbit_hip_sketch_device_seed(int dist, int dtype, const void *m, size_t rows, size_t features, size_t ld, size_t proj,
                                  const uint64_t *seed_device, double scale, void *out, void *workspace, size_t workspace_bytes,
                                  void *stream);
int fewbit_hip_sketch_next_seed(uint64_t *counter_device, uint64_t base, uint64_t *seed_device, void *stream);
uint64_t fewbit_hip_sketch_mix_seed(uint64_t base, uint64_t count);    /* host evaluation of the same function */
/* S[row0 .. row0+nrows) x [col0 .. col0+ncols) itself as fp32, row-major (rounded as the product kernel rounds its operand for
 * `dtype`) -- test seam and debugging aid; the product path never calls it (its own S is in registers or in fragment order) */
int fewbit_hip_sketch_matrix(int dist, int dtype, uint64_t seed, size_t row0, size_t col0, size_t nrows, size_t ncols, float *out,
                             void *stream);
/* launch shape a fewbit_hip_sketch call would use, as JSON: {"kernel", "grid": [x, y, z], "threads", "k_slice", ...} */
int fewbit_hip_sketch_describe(int dist, int dtype, size_t rows, size_t features, size_t proj, char *buf, size_t len);
/* ------------------------------------------------------------------------------------------------------------------
 * Sampled cosine transform of the randomized linear layers:  out[j][:] = scale * DCT-II_ortho(M along its rows)[idx[j]][:]
 *   replaces  `dct(input_view, dim=0, norm='ortho')[proj, ...]`, fewbit/functional/linear.py:113-122 (forward) and :174-183
 *             (backward), with dct = fewbit/fft.py:10-43 -- a full fp32 transform through the FFT library, then a gather
 *   m    rows x features, row-major with leading dimension `ld` (elements), dtype F32 / F16 / BF16; rows = 2^k in [256, 262144] or
 *        3 x 2^k in [768, 49152] or 5 x 2^k in [1280, 40960] (anything else: FEWBIT_ERR_UNSUPPORTED, and
 *        fewbit_hip_sampled_dct_workspace returns 0 -- the caller keeps the library formulation for those); arithmetic and the
 *        intermediate are fp32 whatever the dtype
 *   idx  proj row numbers in [0, rows) as int64 in DEVICE memory (drawn with replacement: duplicates are served one by one)
 *   out  proj x features, contiguous, the dtype of m (fully written)
 *   workspace  fewbit_hip_sampled_dct_workspace(...) = ceil(features / 64) * rows * 256 (the fp32 intermediate) + 2048 + 8 * proj rounded up to
 *              16 (the samples sorted by residue class) bytes, 16-byte aligned; contents are scratch
 * Two launches on `stream` (fewbit_amd/csrc/fewbit_dct.hip); deterministic. */
size_t fewbit_hip_sampled_dct_workspace(int dtype, size_t rows, size_t features, size_t proj);
int fewbit_hip_sampled_dct(int dtype, const void *m, size_t rows, size_t features, size_t ld, const int64_t *idx, size_t proj, double scale,
                           void *out, void *workspace, size_t workspace_bytes, void *stream);
/* The same with the sampled rows a FUNCTION of a 64-bit seed -- no array of row numbers, no launch that draws one (the reference draws
 * `T.multinomial` of uniform probabilities with replacement per call, fewbit/functional/linear.py:114-119, and draws it again in
 * backward from the generator state it saved, :105,153,159-160,176-181):
 *     rows = 2^k <= 2^16:  idx[j] = 16-bit half j % 8 of Philox4x32-10(counter = (j / 8, 0, 0, 3), key = (seed low, seed high))  mod  rows
 *                          (half h = bits 16 (h % 2) .. 16 (h % 2) + 15 of output word h / 2)
 *     any other rows:      idx[j] = (output word j % 4 of Philox4x32-10(counter = (j / 4, 0, 0, 3), key)  x  rows)  >>  32
 * uniform (the second form up to rows / 2^32) with replacement like that draw; forward and
 * backward pass the same seed and sample the same rows.  One workgroup of the first launch evaluates the function (and sorts the samples
 * by residue class for the second launch -- as it does with an explicit idx).
 * seed_device != NULL: the seed is read from that 8-byte aligned DEVICE word when the kernel runs (`seed` is ignored) -- a launch
 * recorded in a hipGraph then draws fresh rows on every replay, fed by fewbit_hip_sketch_next_seed exactly like
 * fewbit_hip_sketch_device_seed.  fewbit_hip_sampled_rows: the function on the HOST (idx: proj int64 in host memory; what tests and
 * a caller that wants the row numbers call). */
int fewbit_hip_sampled_dct_seeded(int dtype, const void *m, size_t rows, size_t features, size_t ld, uint64_t seed, const uint64_t *seed_device,
                                  size_t proj, double scale, void *out, void *workspace, size_t workspace_bytes, void *stream);
int fewbit_hip_sampled_rows(uint64_t seed, size_t rows, size_t proj, int64_t *idx);
/* Philox4x32-10 on the HOST (the generator behind S; known-answer tests run it without a GPU) */
void fewbit_hip_philox4x32(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);
/* xoshiro128++ 1.0 on the HOST (Blackman & Vigna; the stream generator of the Gaussian sketch, seeded by a Philox call per
 * 256-row block): advances `state` by n steps and writes the n outputs */
void fewbit_hip_xoshiro128pp(uint32_t state[4], uint32_t *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* FEWBIT_HIP_H_ */
