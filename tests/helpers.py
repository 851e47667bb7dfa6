"""Shared helpers for the parity tests (bit-exact views, fixture decoding, ULP distance)."""
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / 'tests' / 'golden'
DTYPES = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}


def from_raw(a: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    """Inverse of gen_golden.raw(): uint16 words -> bf16/fp16 tensor, float32 stays."""
    if dtype in (torch.bfloat16, torch.float16):
        return torch.from_numpy(a.astype(np.uint16).view(np.int16).copy()).view(dtype)
    return torch.from_numpy(a.astype(np.float32).copy())


def bits(t: torch.Tensor) -> torch.Tensor:
    """Integer view for bit-exact comparison."""
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.float32:
        return t.view(torch.int32)
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16)
    return t


def assert_bit_equal(a: torch.Tensor, b: torch.Tensor, what: str = '', nan_equal: bool = True):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    ia, ib = bits(a), bits(b)
    neq = ia != ib
    if nan_equal and a.is_floating_point():
        neq &= ~(torch.isnan(a) & torch.isnan(b))        # NaN payload/sign is not part of parity
    if neq.any():
        idx = neq.flatten().nonzero().flatten()[:8]
        raise AssertionError(f'{what}: {int(neq.sum())} of {a.numel()} differ, first at {idx.tolist()}: '
                             f'{a.flatten()[idx].tolist()} vs {b.flatten()[idx].tolist()}')


def ordered(t: torch.Tensor) -> torch.Tensor:
    """Map floats to integers so that adjacent representable values differ by 1."""
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.float32:
        i = t.view(torch.int32).to(torch.int64)
        return torch.where(i < 0, -(i & 0x7fffffff), i)
    i = t.view(torch.int16).to(torch.int64)
    return torch.where(i < 0, -(i & 0x7fff), i)


def ulp_distance(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """|a-b| in units of representable steps of the common dtype (NaN==NaN -> 0)."""
    d = (ordered(a) - ordered(b)).abs()
    both_nan = torch.isnan(a.detach().cpu()) & torch.isnan(b.detach().cpu())
    return torch.where(both_nan, torch.zeros_like(d), d)


def load_tables() -> dict:
    """The built-in quantization tables as float64 numpy arrays (the product's own data file)."""
    with np.load(ROOT / 'fewbit_amd' / 'data' / 'builtin.npz') as z:
        return {k: z[k].copy() for k in z.files}


def forward_value_ok(x: torch.Tensor, y: torch.Tensor, y_ref: torch.Tensor, tight: bool = False) -> torch.Tensor:
    """Element-wise verdict for forward activation values (the only floating-point parity in the path).

    The reference's forward values are ATen's (third-party: MKL vsCdfNorm for contiguous fp32, Sleef erf for
    16-bit and strided inputs -- the two already disagree with each other by up to 5 fp32 steps for x>0 and by
    the full cancellation noise of 1+erf(x/sqrt2) for x<0).  So the bar is stated on the shared formula
    y = x*0.5*(1+erf(x*sqrt(1/2))) evaluated in fp32:
      loose (vs ATen output):   |dy| <= max(1 step of the output dtype at y_ref, 2^-21 * |x|)
      tight (vs the formula with a correctly rounded erf): |dy| <= max(1 step, 2^-24 * |x|)
    2^-24*|x| is what ONE fp32 rounding step of erf costs after the *0.5x scaling.  Non-finite x are excluded
    (ATen itself returns NaN for +inf on some paths); NaN must map to NaN.
    """
    xf, yf, rf = x.detach().cpu().double(), y.detach().cpu().double(), y_ref.detach().cpu().double()
    step_ok = ulp_distance(y, y_ref) <= 1
    scale = 2.0**-24 if tight else 2.0**-21
    abs_ok = (yf - rf).abs() <= scale * xf.abs()
    nan_ok = torch.isnan(xf) & torch.isnan(yf)
    skip = torch.isinf(xf)
    return step_ok | abs_ok | nan_ok | skip


# Input data of the reference's CUDA smoke test (fewbit/cuda/codec_test.cu: TestCodecBlock :62-64, TestGelu :93-98).
# The reference only prints the results; the two vectors pin each other: with the built-in 3-bit GELU table the 16
# inputs fall into exactly the 16 codes of the codec smoke test.
REF_SMOKE_CODES = (6, 5, 6, 1, 7, 0, 4, 2, 2, 3, 0, 4, 5, 5, 6, 7)
REF_SMOKE_GELU_INPUTS = (2.29811567e+00, 6.10855860e-01, 2.29811567e+00, -8.11248159e-01, 9.99900000e+02, -2.49798704e+00,
                         2.26182064e-01, -4.26290283e-01, -4.26290283e-01, -1.00155338e-01, -2.49798704e+00, 2.26182064e-01,
                         6.10855860e-01, 6.10855860e-01, 2.29811567e+00, 9.99900000e+02)


# ---- full-size inputs of the BASELINE configs (shared by tests/golden/gen_golden.py, which runs the REFERENCE on them and
# commits SHA-256 digests, and by tests/test_gpu_parity.py, which regenerates them on the GPU box) -------------------------
FULL_SIZE_CASES = {
    # name: (function, bits, dtype key, rows, cols)
    'c2_gelu3_bf16_4096x4096': ('gelu', 3, 'bf16', 4096, 4096),
    'c3_silu2_f16_8192x8192': ('silu', 2, 'f16', 8192, 8192),
    'c3_silu4_f16_8192x8192': ('silu', 4, 'f16', 8192, 8192),
    'c4shard_gelu3_bf16_8192x4096': ('gelu', 3, 'bf16', 8192, 4096),
}


def border_specials(inner_borders: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """NaN (both signs), +-inf, +-0, every border and its two neighbours in `dtype`, +-100, +-200."""
    b = inner_borders.to(dtype)
    it = torch.int32 if dtype == torch.float32 else torch.int16
    nb = b.view(it)
    fixed = torch.tensor([float('nan'), float('inf'), -float('inf'), 0.0, -0.0, 100.0, -100.0, 200.0, -200.0, 1e-30, -1e-30]).to(dtype)
    neg_nan = torch.tensor([-1], dtype=it).view(dtype)
    return torch.cat([fixed, b, (nb + 1).view(dtype), (nb - 1).view(dtype), neg_nan])


def full_size_inputs(case: str, tables: dict):
    """(x, gy, inner borders, levels) of a BASELINE-size case: seeded host randn cast to the dtype, with the
    special values spliced in at the start and at the end of x."""
    name, bits, dt, rows, cols = FULL_SIZE_CASES[case]
    dtype = DTYPES[dt]
    borders = torch.tensor(tables[f'{name}{bits:02d}-borders']).to(dtype)[1:-1].contiguous()
    levels = torch.tensor(tables[f'{name}{bits:02d}-levels']).to(dtype)
    x = torch.randn(rows * cols, generator=torch.Generator().manual_seed(0)).to(dtype)
    gy = torch.randn(rows * cols, generator=torch.Generator().manual_seed(1)).to(dtype)
    sp = border_specials(borders, dtype)
    x[:sp.numel()] = sp
    x[-sp.numel():] = sp
    return x, gy, borders, levels


# ---- BASELINE config 4 in full: 4 tensors of 16384x4096 bf16, each cut in two halves -> 8 shards, tensor t half h on GPU 2t+h
# (SURVEY 8(e)).  The reference runs on each WHOLE tensor (tests/golden/gen_golden.py c4); a shard's expected state / gx is the
# slice of that run at the shard's offsets, which is exactly the claim "sharding == slicing the unsharded result".
C4_TENSORS, C4_ROWS, C4_COLS, C4_BITS = 4, 16384, 4096, 3


def c4_tensor_inputs(t: int, tables: dict):
    """(x, gy, inner borders, levels) of whole tensor `t` of config 4: seeded host randn, the special values spliced in at
    both ends AND on both sides of the cut between its two shards."""
    dtype = torch.bfloat16
    borders = torch.tensor(tables['gelu03-borders']).to(dtype)[1:-1].contiguous()
    levels = torch.tensor(tables['gelu03-levels']).to(dtype)
    n = C4_ROWS * C4_COLS
    x = torch.randn(n, generator=torch.Generator().manual_seed(400 + t)).to(dtype)
    gy = torch.randn(n, generator=torch.Generator().manual_seed(500 + t)).to(dtype)
    sp = border_specials(borders, dtype)
    x[:sp.numel()] = sp
    x[-sp.numel():] = sp
    x[n // 2 - sp.numel():n // 2] = sp
    x[n // 2:n // 2 + sp.numel()] = sp
    return x, gy, borders, levels


def c4_shard(rank: int):
    """(tensor index, element range, state byte range) of GPU `rank`'s shard of config 4"""
    # (sharding.py loaded by path, not through the package: tests/golden/gen_golden.py calls this with the REFERENCE's
    # operator library loaded, which registers the same torch.ops namespace as the package's own)
    import importlib.util
    spec = importlib.util.spec_from_file_location('_fewbit_sharding', ROOT / 'fewbit_amd' / 'sharding.py')
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    shard_range, state_range = sharding.shard_range, sharding.state_range
    t, h = divmod(rank, 2)
    begin, end = shard_range(C4_ROWS * C4_COLS, 2, h)
    return t, (begin, end), state_range(begin, end, C4_BITS)


def sha256_of(t: torch.Tensor) -> str:
    import hashlib
    t = t.detach().cpu().contiguous()
    return hashlib.sha256(t.view(torch.uint8).numpy().tobytes()).hexdigest()
