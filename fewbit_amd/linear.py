"""Linear layers that keep a *compressed* copy of their input for the weight gradient.

Counterpart of the reference's randomized layers (``fewbit/functional/linear.py``, ``fewbit/modules/linear.py``;
paper: "Memory-Efficient Backpropagation through Large Linear Layers", arXiv:2201.13195).  Same public names
(``linear_crs``, ``linear_grp``, ``linear_randomized``; ``LinearCRS``, ``LinearGRP``, ``RandomizedLinear``) and
constructor / call signatures; the implementation is this repository's own:

* forward is exactly ``F.linear``; the input gradient is exact; only ``dL/dW = G^T X`` (G, X: rows x features) is
  estimated, from a sketch ``S`` (p x rows, ``E[S^T S] = I``): saved for backward is ``S X`` (p rows instead of
  ``rows``), backward recomputes ``S`` from the saved generator state and forms ``(S G)^T (S X)``;
* the sketch is drawn in the *input's dtype*, so with bf16/fp16 activations both sketch GEMMs run on the matrix
  cores (hipBLASLt through ``torch.matmul``) -- the reference draws fp32 and cannot multiply it with 16-bit inputs;
* ``sketch_dtype`` (extension): dtype the two dense sketch products ``S X`` and ``S G`` are computed in -- and, on the
  GPU kernel, the dtype the projection is KEPT in.  ``S G`` costs four times the exact weight-gradient GEMM at ratio 0.2, so
  for fp32 layers ``sketch_dtype=torch.bfloat16`` moves those flops onto the bf16 matrix cores; with this package's kernel
  (which rounds fp32 operands to bf16 anyway) what the option adds is a bf16 projection -- half the saved bytes -- and a
  bf16 final GEMM: RoBERTa-base fp32, Rademacher, ratio 0.2: 1.01x the vanilla step at -26.4 % peak memory instead of
  1.02x at -21.5 %.  The rounding of the projections to 8 bits is zero-mean and far below the variance of the estimator;
* every sketch is unbiased.  The reference's ``'dct'``/``'dft'`` branches scale the sampled rows by ``p * rows``
  (``fewbit/functional/linear.py:124-137``) where unbiasedness needs ``rows / p``, and its ``'dft'`` backward drops
  the imaginary part before the product (:189-197, :214-216); neither defect is reproduced (they are not covered by
  the reference's tests, which only run the default ``'gaussian'``, ``fewbit/modules/linear_test.py``);
* without a user generator the sketch seed comes from the host default generator (so ``torch.manual_seed`` makes
  runs reproducible) and never reads back from the device -- no stream synchronisation in forward or backward;
* on the GPU the dense sketches (``'gaussian'``, ``'rademacher'``) of fp32 / fp16 / bf16 tensors run on this package's own
  gfx950 kernels (``fewbit_amd/csrc/fewbit_sketch.hip`` through the C-ABI ``fewbit_hip_sketch``): ``S`` is a pure function
  of a 64-bit seed (Philox4x32-10; Gaussian: xoshiro128++ streams seeded by it) -- Rademacher signs are generated in registers
  and fed straight to the matrix cores, a Gaussian ``S`` of a layer wider than 256 features is written once per product into a
  scratch workspace as MFMA fragments and read back (it costs as much to generate as to multiply: regenerating it per column
  tile was slower than ``randn`` + ``matmul``); what a layer keeps for backward is the ``p x features`` projection and ONE
  integer, and backward regenerates the same ``S`` from that integer.  The reference draws ``proj x rows`` random numbers into
  device memory twice per layer and step (fewbit/functional/linear.py:133-137,195-199).  fp32 operands are rounded to bf16 on their
  way into the matrix pipe (accumulation is fp32): a zero-mean relative perturbation of 2^-9 per element under an estimator
  whose own relative noise is ~ sqrt(rows / p).  ``use_native_sketch(False)`` (or ``FEWBIT_SKETCH_NATIVE=0``) selects
  the PyTorch formulation (randn / randint + matmul) instead, and so does an EXPLICIT ``sketch_dtype=torch.float32`` /
  ``float64`` (a request for products of that precision); host tensors, float64 and 'dft' always take it;
* the native path can be captured into a hipGraph (``torch.cuda.graph``) after one eager warm-up call per device: while
  the stream is capturing, the seed is a device word that a recorded one-thread kernel re-derives on every replay
  (``_sketch_seed``), so a replayed training step draws a fresh ``S`` each time -- a seed recorded by value would repeat
  one matrix for ever.  (The reference reads the generator state back in forward and cannot be captured.)

The sampled transforms: 'dct' on 2-D GPU tensors of 2^8 .. 2^18, 3 x 2^8 .. 3 x 2^14 or 5 x 2^8 .. 5 x 2^13 rows runs on this package's kernel pair (``fewbit_hip_sampled_dct``,
``fewbit_amd/csrc/fewbit_dct.hip``: a four-step fp32 FFT in LDS that writes only the sampled rows -- the torch.fft formulation costs
110-120 x the bytes of the result, profiles/r06_sketch_bench.json); other shapes and 'dft' are PyTorch-level code.  Inside the layer the
sampled rows are, like ``S``, a function of the call's 64-bit seed that the kernel evaluates itself (``fewbit_hip_sampled_dct_seeded``;
``cabi.sampled_rows(seed, rows, p)`` is the same function on the host): no ``randint`` launch, no saved RNG state, and the layer can be
captured into a hipGraph.  ``use_native_sketch(False)`` selects torch.fft + randint.  SURVEY section 8f, row 4.
"""
import contextlib
import os
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .fft import dct

__all__ = ('MATMUL_TYPES', 'projection_dim', 'linear_crs', 'linear_grp', 'linear_randomized', 'LinearCRS', 'LinearGRP',
           'RandomizedLinear', 'use_native_sketch', 'sampled_transform', 'sampled_transform_path')

MATMUL_TYPES = ('dct', 'dft', 'gaussian', 'rademacher')


def projection_dim(rows: int, proj_dim_ratio: Optional[float] = None, proj_dim: Optional[int] = None,
                   proj_dim_max: Optional[int] = None, proj_dim_min: Optional[int] = None) -> int:
    """Number of sketch rows for an input of ``rows`` rows: ``proj_dim``, else ``int(ratio * rows)``, clamped to
    ``[proj_dim_min, proj_dim_max]`` (reference: ``LinearGRPFunc.calc_proj_dim``, fewbit/functional/linear.py:73-83)."""
    if proj_dim:
        p = int(proj_dim)
    elif proj_dim_ratio:
        p = int(proj_dim_ratio * rows)
    else:
        p = int(rows)
    if proj_dim_min:
        p = max(int(proj_dim_min), p)
    if proj_dim_max:
        p = min(int(proj_dim_max), p)
    return max(p, 1)


# ---- random state that can be replayed in backward ------------------------------------------------------------------

def _capture_rng(generator: Optional[torch.Generator], device: torch.device):
    """-> (token, generator to draw from now).  The token rebuilds an identical generator later."""
    if generator is not None:
        return ('state', generator.device, generator.get_state()), generator
    seed = int(torch.randint(0, 2**62, (), dtype=torch.int64).item())          # host generator: no device round trip
    gen_device = device if device.type == 'cuda' else torch.device('cpu')
    return ('seed', gen_device, seed), torch.Generator(device=gen_device).manual_seed(seed)


def _replay_rng(token) -> torch.Generator:
    kind, device, payload = token
    gen = torch.Generator(device=device)
    if kind == 'seed':
        gen.manual_seed(payload)
    else:
        gen.set_state(payload)
    return gen


# ---- the gfx950 kernel for the dense sketches ---------------------------------------------------------------------

_NATIVE_SKETCH = os.environ.get('FEWBIT_SKETCH_NATIVE', '1') not in ('0', 'no', 'false')


def use_native_sketch(on: Optional[bool] = None) -> bool:
    """Query / set whether GPU tensors take the Philox-in-register MFMA kernel for 'gaussian' / 'rademacher' sketches
    (default) or the PyTorch formulation.  Returns the previous setting."""
    global _NATIVE_SKETCH
    prev = _NATIVE_SKETCH
    if on is not None:
        _NATIVE_SKETCH = bool(on)
    return prev


def _native_sketch_applies(kind: str, mat: torch.Tensor, sketch_dtype) -> bool:
    """The kernel multiplies on the 16-bit matrix pipe (fp32 input is rounded to bf16, sums are fp32).  ``sketch_dtype=None``
    (the default) and the 16-bit dtypes take it; an EXPLICIT ``sketch_dtype=torch.float32`` / ``float64`` asks for products of
    that precision and gets the PyTorch formulation (S drawn into memory, library GEMM in that dtype) instead."""
    return (_NATIVE_SKETCH and _INJECTED is None and kind in ('gaussian', 'rademacher') and mat.device.type == 'cuda'
            and mat.dtype in (torch.float32, torch.float16, torch.bfloat16) and mat.dim() == 2 and mat.shape[0] > 0
            and sketch_dtype in (None, torch.bfloat16, torch.float16))


def _mix64(a: int, b: int) -> int:
    """splitmix64 of (a, b): seed of one call from a device generator's (seed, offset) without touching the device"""
    x = (a * 0x9E3779B97F4A7C15 + b + 0x632BE59BD9B4E019) & 0xffffffffffffffff
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xffffffffffffffff
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xffffffffffffffff
    return x ^ (x >> 31)


def _draw_seed(generator: Optional[torch.Generator]) -> int:
    """64-bit seed of one sketch.  Host generators (the default one included) are advanced by one draw; a device generator
    is advanced by bumping its Philox offset -- neither reads back from the device."""
    if generator is None or generator.device.type == 'cpu':
        return int(torch.randint(0, 2**62, (), dtype=torch.int64, generator=generator).item())
    offset = generator.get_offset()
    generator.set_offset(offset + 4)
    return _mix64(generator.initial_seed(), offset)


_REPLAY_COUNTERS = {}


def _replay_counter(device: torch.device) -> torch.Tensor:
    """One int64 word per device that the recorded seed kernels advance.  It must exist BEFORE a capture starts (a tensor
    allocated while capturing belongs to the graph's pool and its zero-fill would be replayed too): every eager call of the
    native path makes sure it does, so the customary warm-up run before `torch.cuda.graph(...)` is enough."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    counter = _REPLAY_COUNTERS.get(key)
    if counter is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('fewbit.linear_grp: run the layer once on this device before capturing it into a graph '
                               '(the per-device replay counter of the sketch seeds cannot be created during capture)')
        counter = _REPLAY_COUNTERS[key] = torch.zeros(1, dtype=torch.int64, device=device)
    return counter


_WARNED_CAPTURE_GENERATOR = False


def _warn_device_generator_in_capture() -> None:
    global _WARNED_CAPTURE_GENERATOR
    if not _WARNED_CAPTURE_GENERATOR:
        _WARNED_CAPTURE_GENERATOR = True
        import warnings
        warnings.warn('fewbit.linear_grp: a device generator passed while the stream is being captured into a graph only contributes its '
                      'initial seed -- its offset is not advanced, and replays draw from (that seed, the per-device replay counter); '
                      'eager and captured runs with the same generator therefore see different sketches', RuntimeWarning, stacklevel=4)


def _sketch_seed(generator: Optional[torch.Generator], device: torch.device):
    """Seed of one sketch: an int drawn on the host -- or, while the stream is being captured into a hipGraph, a device word
    that a recorded kernel re-derives from (a host draw made at capture time, the replay counter) on every replay, so that a
    replayed training step meets a fresh S each time instead of the one matrix whose seed was recorded."""
    counter = _replay_counter(device)
    if not torch.cuda.is_current_stream_capturing():
        return _draw_seed(generator)
    from . import cabi
    # (while capturing, a DEVICE generator's offset is not touched: reading or moving it belongs to the graph machinery of
    # torch.cuda; a host generator is used as given.  The counter is one word per device shared by every captured graph: graphs replayed one after the other
    # see consecutive counts (reproducible), graphs replayed CONCURRENTLY on several streams still get distinct counts -- the
    # bump is an atomic add -- but which graph gets which is then up to the hardware.)
    if generator is not None and generator.device.type != 'cpu':
        # the user's generator still determines the stream: the base of the recorded seed kernel is derived from its initial seed
        # (a host-side read; its offset is neither read nor moved) and the per-device replay counter, so two captures with
        # generators of different seeds draw different matrices, and the host's global RNG state is left alone
        _warn_device_generator_in_capture()
        return cabi.next_sketch_seed(counter, _mix64(generator.initial_seed(), 0x6361707475726564))
    return cabi.next_sketch_seed(counter, _draw_seed(generator))


def _native_sketch(kind: str, mat: torch.Tensor, p: int, seed, scale: float) -> torch.Tensor:
    from . import cabi
    if mat.stride(1) != 1 or mat.stride(0) < mat.shape[1]:      # (e.g. the expanded gradient of a sum: strides (0, 0))
        mat = mat.contiguous()
    return cabi.sketch(kind, mat, p, seed, scale)


def _native_dct(mat: torch.Tensor, p: int, seed, scale: float) -> torch.Tensor:
    """``scale * dct(mat, dim=0, norm='ortho')[rows(seed)]`` on this package's kernel pair: M is read once, one fp32 intermediate goes
    out and back, only the p sampled rows are written (the torch formulation materialises the whole transform in fp32 first)"""
    from . import cabi
    if mat.stride(1) != 1 or mat.stride(0) < mat.shape[1]:
        mat = mat.contiguous()
    return cabi.sampled_dct_seeded(mat, p, seed, scale)


_INJECTED: Optional[torch.Tensor] = None


@contextlib.contextmanager
def inject_sketch(S: torch.Tensor):
    """Checker hook: inside the block every dense sketch (forward AND backward of ``linear_grp`` with ``'gaussian'`` /
    ``'rademacher'``) is the given ``p x rows`` matrix instead of a fresh draw.  The fixture tests inject the matrix the
    reference drew (tests/golden/linear_draw_ref.npz) and compare the weight gradient with the reference's own."""
    global _INJECTED
    prev, _INJECTED = _INJECTED, S
    try:
        yield
    finally:
        _INJECTED = prev


def _dense_sketch(kind: str, p: int, rows: int, like: torch.Tensor, gen: torch.Generator,
                  draw_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    if _INJECTED is not None:
        if tuple(_INJECTED.shape) != (p, rows):
            raise ValueError(f'injected sketch is {tuple(_INJECTED.shape)}, this call needs {(p, rows)}')
        return _INJECTED.to(like.device, like.dtype)
    if kind == 'gaussian':
        # drawn in ONE dtype (`draw_dtype`, recorded by the forward) and then cast to the operand: randn's stream depends
        # on the dtype, so a backward whose grad_output has another dtype than the forward's input (autocast) would
        # otherwise replay a different S and return pure noise
        return torch.randn((p, rows), generator=gen, device=gen.device, dtype=draw_dtype or like.dtype).to(like.device, like.dtype)
    signs = torch.randint(0, 2, (p, rows), generator=gen, device=gen.device, dtype=torch.int8).to(like.device)
    return (signs.to(like.dtype) * 2) - 1                                       # Rademacher: +-1


def _sampled_rows(p: int, rows: int, like: torch.Tensor, gen: torch.Generator) -> torch.Tensor:
    return torch.randint(0, rows, (p, ), generator=gen, device=gen.device).to(like.device)


def _native_dct_applies(mat: torch.Tensor) -> bool:
    """The gfx950 sampled-DCT kernel pair (fewbit_amd/csrc/fewbit_dct.hip) takes 2-D fp32 / fp16 / bf16 GPU tensors whose row count is
    2^k in [256, 262144] (RoBERTa's 128 x 128 tokens = 16384) or 3 x 2^k in [768, 49152] (32 sequences of 384 tokens = 12288) or 5 x 2^k in [1280, 40960]; everything
    else keeps the torch.fft formulation."""
    if not (_NATIVE_SKETCH and mat.device.type == 'cuda' and mat.dim() == 2 and mat.dtype in (torch.float32, torch.float16, torch.bfloat16)):
        return False
    rows = mat.shape[0]
    return _dct_rows_supported(rows) and mat.shape[1] > 0


def _dct_rows_supported(rows: int) -> bool:
    """2^k rows, k = 8 .. 18, 3 x 2^k rows, k = 8 .. 14, or 5 x 2^k rows, k = 8 .. 13 (fewbit_dct.hip::split_rows)"""
    three = rows % 3 == 0
    five = not three and rows % 5 == 0
    two = rows // 3 if three else rows // 5 if five else rows
    return 256 <= two <= (16384 if three else 8192 if five else 262144) and two & (two - 1) == 0


def sampled_transform_path(kind: str, mat: torch.Tensor) -> str:
    """Which code computes the sampled transform ``kind`` ('dct' / 'dft') of ``mat`` (what bench.py prints beside its time)."""
    if kind == 'dct' and _native_dct_applies(mat):
        return 'gfx950 kernel pair fewbit_hip_sampled_dct (four-step fp32 FFT in LDS, only the sampled rows are written; in the layer: rows of a seed)'
    return 'torch.fft (rocFFT on the GPU): full transform along dim 0 in fp32, then the gather of the sampled rows'


def sampled_transform(kind: str, mat: torch.Tensor, p: int, gen: torch.Generator, seed: int = 1234, scale: float = 1.0) -> torch.Tensor:
    """One estimator product of the layer, ``scale * transform(mat)[p sampled rows]``, on the path ``linear_grp`` takes for this
    ``kind`` and ``mat`` (what bench.py and tools/ time): the kernel pair with rows of ``seed``, or torch.fft + randint from ``gen``."""
    if kind == 'dct' and _native_dct_applies(mat):
        return _native_dct(mat, p, seed, scale)
    return _sketch(kind, mat, p, gen, scale=scale)


def _sketch(kind: str, mat: torch.Tensor, p: int, gen: torch.Generator, sketch_dtype=None, draw_dtype=None, scale: float = 1.0) -> torch.Tensor:
    """``scale * S @ mat`` (``E[S^T S] = p * I`` for the dense sketches, ``(p / rows) * I`` for the sampled transforms)."""
    rows = mat.shape[0]
    if kind in ('gaussian', 'rademacher'):
        if sketch_dtype is not None and sketch_dtype != mat.dtype:
            low = mat.to(sketch_dtype)
            out = (_dense_sketch(kind, p, rows, low, gen, draw_dtype) @ low).to(mat.dtype)
        else:
            out = _dense_sketch(kind, p, rows, mat, gen, draw_dtype) @ mat
        return out if scale == 1.0 else out * scale
    idx = _sampled_rows(p, rows, mat, gen)
    if kind == 'dct':
        out = dct(mat, dim=0, norm='ortho')[idx]
        return out if scale == 1.0 else out * scale
    work = mat if mat.dtype in (torch.float32, torch.float64) else mat.float()
    out = torch.fft.fft(work, dim=0, norm='ortho')[idx]                         # complex
    return out if scale == 1.0 else out * scale


class _LinearGRP(torch.autograd.Function):

    @staticmethod
    def forward(ctx, input, weight, bias, p: int, kind: str, generator, sketch_dtype=None):
        flat = input.reshape(-1, input.shape[-1])
        rows = flat.shape[0]
        ctx.native_seed = None
        if _native_sketch_applies(kind, flat, sketch_dtype):
            # S lives nowhere: the projection and the seed are all that is kept
            ctx.native_seed = _sketch_seed(generator, flat.device)
            # sketch_dtype (16-bit) for a wider input: the projection is computed from, and KEPT in, that dtype -- half the
            # saved bytes of an fp32 layer, and the small GEMM of backward runs on the 16-bit matrix pipe as well
            low = sketch_dtype if sketch_dtype is not None and sketch_dtype != flat.dtype else None
            ctx.low = low
            # (the sketch before or after the layer's own GEMM: no difference, profiles/r05_roberta_ab_order.txt)
            sketch = _native_sketch(kind, flat.detach() if low is None else flat.detach().to(low), p, ctx.native_seed, 1.0 / p)
            ctx.save_for_backward(sketch, weight)
            ctx.p, ctx.kind = p, kind
            ctx.has_bias = bias is not None
            return F.linear(input, weight, bias)
        if kind == 'dct' and _native_dct_applies(flat):
            # the sampled rows live nowhere either: a function of the seed that the kernel evaluates itself (no randint launch, no
            # RNG state to save and replay; while a graph is being captured the seed is a device word, like the dense sketches')
            ctx.native_seed = _sketch_seed(generator, flat.device)
            ctx.low = None
            sketch = _native_dct(flat.detach(), p, ctx.native_seed, rows / p)
            ctx.save_for_backward(sketch, weight)
            ctx.p, ctx.kind = p, kind
            ctx.has_bias = bias is not None
            return F.linear(input, weight, bias)
        token, gen = _capture_rng(generator, input.device)
        scale = 1.0 / p if kind in ('gaussian', 'rademacher') else rows / p
        draw_dtype = sketch_dtype or flat.dtype
        sketch = _sketch(kind, flat.detach(), p, gen, sketch_dtype, draw_dtype, scale)
        ctx.save_for_backward(sketch, weight)
        ctx.token, ctx.p, ctx.kind, ctx.sketch_dtype, ctx.draw_dtype = token, p, kind, sketch_dtype, draw_dtype
        ctx.has_bias = bias is not None
        return F.linear(input, weight, bias)

    @staticmethod
    def backward(ctx, grad_output):
        sketch, weight = ctx.saved_tensors
        grad_input = grad_weight = grad_bias = None
        if ctx.needs_input_grad[0]:
            grad_input = grad_output @ weight
        flat = grad_output.reshape(-1, grad_output.shape[-1])
        if ctx.needs_input_grad[1] and ctx.native_seed is not None:
            # the same S again, from the same seed (a grad_output of another dtype than the forward's input -- autocast --
            # still meets the same matrix: S does not depend on the operand dtype beyond its final rounding)
            g2 = flat if flat.dtype in (torch.float32, torch.float16, torch.bfloat16) else flat.float()
            if ctx.low is not None:
                g2 = g2.to(ctx.low)
            if ctx.kind == 'dct':
                proj = _native_dct(g2, ctx.p, ctx.native_seed, 1.0)
            else:
                proj = _native_sketch(ctx.kind, g2, ctx.p, ctx.native_seed, 1.0)
            grad_weight = (proj.to(sketch.dtype).T @ sketch).to(weight.dtype)
        elif ctx.needs_input_grad[1]:
            proj = _sketch(ctx.kind, flat, ctx.p, _replay_rng(ctx.token), ctx.sketch_dtype, ctx.draw_dtype)
            if proj.is_complex():                                               # Re((F G)^H (F X))
                grad_weight = (proj.real.T @ sketch.real + proj.imag.T @ sketch.imag).to(weight.dtype)
            else:
                grad_weight = (proj.T @ sketch).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = flat.sum(dim=0)
        return grad_input, grad_weight, grad_bias, None, None, None, None


def linear_grp(input: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
               proj_dim_ratio: Optional[float] = None, proj_dim: Optional[int] = None,
               proj_dim_max: Optional[int] = None, proj_dim_min: Optional[int] = None, matmul: str = 'gaussian',
               generator: Optional[torch.Generator] = None, sketch_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``F.linear(input, weight, bias)`` whose weight gradient is estimated through a random projection of the rows
    (argument order of the reference's ``linear_grp``, fewbit/functional/linear.py:85-90)."""
    if proj_dim_ratio is None and proj_dim is None:
        raise ValueError('Either proj_dim or proj_dim_ratio should be specified.')
    if proj_dim_min is not None and proj_dim_min <= 0:
        raise ValueError('Param proj_dim_min should be strictly positive.')
    if proj_dim_min and proj_dim_max and proj_dim_max < proj_dim_min:
        raise ValueError('Param proj_dim_min should be not greater than param proj_dim_max.')
    if matmul not in MATMUL_TYPES:
        raise ValueError(f'Unexpected matmul type: {matmul}.')
    rows = input.numel() // input.shape[-1] if input.numel() else 0
    p = projection_dim(rows, proj_dim_ratio, proj_dim, proj_dim_max, proj_dim_min)
    return _LinearGRP.apply(input, weight, bias, p, matmul, generator, sketch_dtype)


linear_randomized = linear_grp


class _LinearCRS(torch.autograd.Function):
    """Column sampling of the weight gradient: ``nopairs`` draws (with replacement) from the ``in_features`` columns;
    only the drawn columns of the input are kept, each scaled by ``count / (nopairs / in_features)`` so that the
    estimate is unbiased (behaviour of fewbit/functional/linear.py:28-62)."""

    @staticmethod
    def forward(ctx, input, weight, bias, nopairs: int):
        in_features = weight.shape[1]
        draws = torch.randint(0, in_features, (nopairs, ), device=input.device)
        counts = torch.bincount(draws, minlength=in_features)
        scale = counts.to(input.dtype) * (in_features / nopairs)
        flat = input.detach().reshape(-1, in_features)
        # a dense (rows x in_features) product with a mostly-zero scale would save nothing: keep the hit columns only.
        # nonzero() has a data-dependent size and therefore reads back from the device -- inherent to this estimator.
        cols = torch.nonzero(counts, as_tuple=True)[0]
        ctx.save_for_backward(flat[:, cols] * scale[cols], weight, cols)
        ctx.has_bias = bias is not None
        return F.linear(input, weight, bias)

    @staticmethod
    def backward(ctx, grad_output):
        kept, weight, cols = ctx.saved_tensors
        grad_input = grad_weight = grad_bias = None
        if ctx.needs_input_grad[0]:
            grad_input = grad_output @ weight
        flat = grad_output.reshape(-1, grad_output.shape[-1])
        if ctx.needs_input_grad[1]:
            grad_weight = torch.zeros_like(weight)
            grad_weight[:, cols] = (flat.T @ kept).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = flat.sum(dim=0)
        return grad_input, grad_weight, grad_bias, None


def linear_crs(input: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], nopairs: int) -> torch.Tensor:
    """``F.linear`` with a column-sampled weight gradient (reference: ``linear_crs``, fewbit/functional/linear.py:65)."""
    if nopairs < 1:
        raise ValueError('Number of sampled pairs should be strictly positive.')
    return _LinearCRS.apply(input, weight, bias, int(nopairs))


# ---- modules ---------------------------------------------------------------------------------------------------------

class LinearCRS(torch.nn.Linear):
    """:class:`torch.nn.Linear` whose weight gradient is estimated from ``proj_dim`` sampled input columns
    (default ``out_features // 2``, as in fewbit/modules/linear.py:22-36; the reference's constructor forwards
    ``proj_dim`` into the ``bias`` slot of ``nn.Linear`` and its ``extra_repr`` reads an undefined attribute --
    both fixed here)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None,
                 proj_dim: Optional[int] = None) -> None:
        super().__init__(in_features, out_features, bias, device, dtype)
        self.proj_dim: int = proj_dim or max(out_features // 2, 1)

    @property
    def nopairs(self) -> int:
        return self.proj_dim

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return linear_crs(input, self.weight, self.bias, self.proj_dim)

    def extra_repr(self) -> str:
        return f'{super().extra_repr()}, nopairs={self.nopairs}'


class LinearGRP(torch.nn.Linear):
    r"""Drop-in :class:`torch.nn.Linear` (:math:`y = xA^T + b`) that stores a random projection of its input along
    the batch dimension instead of the input itself and estimates the weight gradient from it.

    Parameters (after those of ``nn.Linear``; either ``proj_dim_ratio`` or ``proj_dim`` is required)
    ----------
    proj_dim_ratio : float, optional
        ``proj_dim`` as a fraction of the number of input rows.
    proj_dim : int, optional
        Exact number of rows of the projection.
    proj_dim_min, proj_dim_max : int, optional
        Bounds on the number of rows.
    matmul : {'dct', 'dft', 'gaussian', 'rademacher'}, default='gaussian'
        Kind of random projection.
    generator : torch.Generator, optional
        Source of randomness; without it the host default generator seeds every call.
    sketch_dtype : torch.dtype, optional
        dtype of the dense sketch products (extension; e.g. ``torch.bfloat16`` for an fp32 layer on the GPU).

    Examples:

        >>> m = fewbit.RandomizedLinear(20, 30, proj_dim_ratio=0.5)
        >>> m(torch.randn(128, 20)).size()
        torch.Size([128, 30])
    """

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None,
                 proj_dim_ratio: Optional[float] = None, proj_dim: Optional[int] = None,
                 proj_dim_min: Optional[int] = None, proj_dim_max: Optional[int] = None, matmul: str = 'gaussian',
                 generator: Optional[torch.Generator] = None, sketch_dtype: Optional[torch.dtype] = None) -> None:
        super().__init__(in_features, out_features, bias, device, dtype)
        self.generator = generator
        self.sketch_dtype = sketch_dtype
        self.matmul = matmul
        self.proj_dim_ratio = proj_dim_ratio
        self.proj_dim = proj_dim
        self.proj_dim_max = proj_dim_max
        self.proj_dim_min = proj_dim_min

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return linear_grp(input, self.weight, self.bias, self.proj_dim_ratio, self.proj_dim, self.proj_dim_max,
                          self.proj_dim_min, self.matmul, self.generator, self.sketch_dtype)

    def extra_repr(self) -> str:
        return ', '.join([super().extra_repr(), f'matmul={self.matmul}', f'proj_dim={self.proj_dim}',
                          f'proj_dim_ratio={self.proj_dim_ratio}', f'proj_dim_max={self.proj_dim_max}',
                          f'proj_dim_min={self.proj_dim_min}'])


RandomizedLinear = LinearGRP
