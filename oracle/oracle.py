"""ctypes front-end of oracle/liboracle.so (see fewbit_oracle.c for citations).

TEST INFRASTRUCTURE ONLY: the checker, never the thing measured or shipped.
All functions take/return CPU torch tensors (bf16/fp16 are passed as raw
16-bit words to the C side, which does its own software conversions).
"""
import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np
import torch

__all__ = [
    'CONTINUOUS', 'STEPWISE1', 'build', 'lib', 'bitwidth', 'state_nbytes', 'state_nbytes_padded', 'deflate',
    'inflate', 'searchsorted', 'activation', 'quantize', 'quantize_backward', 'stepwise1_forward',
    'stepwise1_backward', 'convert', 'ref_codec',
]

HERE = Path(__file__).resolve().parent

# ids shared with include/fewbit_hip.h
CONTINUOUS = ('celu', 'elu', 'gelu', 'hardswish', 'logsigmoid', 'mish', 'selu', 'sigmoid', 'silu', 'softplus',
              'softsign', 'tanh', 'tanhshrink', 'identity', 'identity_fold')
STEPWISE1 = ('hardshrink', 'hardsigmoid', 'hardtanh', 'leaky_relu', 'relu', 'relu6', 'softshrink', 'threshold')

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}

_lib = None


def build(ref: bool = False) -> None:
    """Compile liboracle.so (and, when asked and /root/reference exists, oracle/_ref)."""
    subprocess.run(['make', '-s', '-C', str(HERE)], check=True)
    if ref and Path('/root/reference/fewbit/cpu/codec.h').exists():
        subprocess.run(['make', '-s', '-C', str(HERE), 'ref'], check=True)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        # FEWBIT_ORACLE_LIB: the sanitizer build of the same restatement (`make -C oracle asan-test`)
        so = Path(os.environ.get('FEWBIT_ORACLE_LIB') or HERE / 'liboracle.so')
        if not so.exists():
            build()
        L = ctypes.CDLL(str(so))
        vp, sz, i32, dbl = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double
        L.fewbit_oracle_state_nbytes.restype = sz
        L.fewbit_oracle_state_nbytes.argtypes = [sz, i32]
        L.fewbit_oracle_state_nbytes_padded.restype = sz
        L.fewbit_oracle_state_nbytes_padded.argtypes = [sz, i32]
        L.fewbit_oracle_bitwidth.restype = i32
        L.fewbit_oracle_bitwidth.argtypes = [i32]
        L.fewbit_oracle_deflate.restype = None
        L.fewbit_oracle_deflate.argtypes = [vp, sz, vp, i32]
        L.fewbit_oracle_inflate.restype = None
        L.fewbit_oracle_inflate.argtypes = [vp, sz, vp, i32]
        L.fewbit_oracle_searchsorted.restype = None
        L.fewbit_oracle_searchsorted.argtypes = [i32, vp, sz, vp, i32, vp]
        L.fewbit_oracle_activation.restype = None
        L.fewbit_oracle_activation.argtypes = [i32, i32, vp, sz, dbl, dbl, vp]
        L.fewbit_oracle_quantize.restype = i32
        L.fewbit_oracle_quantize.argtypes = [i32, i32, vp, sz, vp, i32, dbl, dbl, vp, vp]
        L.fewbit_oracle_quantize_backward.restype = i32
        L.fewbit_oracle_quantize_backward.argtypes = [i32, vp, sz, vp, vp, i32, vp]
        L.fewbit_oracle_stepwise1_forward.restype = None
        L.fewbit_oracle_stepwise1_forward.argtypes = [i32, i32, vp, sz, dbl, dbl, vp, vp]
        L.fewbit_oracle_stepwise1_backward.restype = None
        L.fewbit_oracle_stepwise1_backward.argtypes = [i32, i32, vp, sz, vp, dbl, vp]
        L.fewbit_oracle_convert.restype = None
        L.fewbit_oracle_convert.argtypes = [vp, i32, vp, i32, sz]
        _lib = L
    return _lib


def _cpu(t: torch.Tensor) -> torch.Tensor:
    assert t.device.type == 'cpu', 'the oracle works on host tensors only'
    return t.contiguous()


def bitwidth(nlevels: int) -> int:
    return lib().fewbit_oracle_bitwidth(nlevels)


def state_nbytes(n: int, k: int) -> int:
    """Reference CPU state length ceil(k*n/8) (fewbit/cpu/gelu.cc:18-20)."""
    return lib().fewbit_oracle_state_nbytes(n, k)


def state_nbytes_padded(n: int, k: int) -> int:
    """Reference GPU state length k*ceil(n/8) (fewbit/cuda/activation.cc:350-351)."""
    return lib().fewbit_oracle_state_nbytes_padded(n, k)


def deflate(codes, k: int) -> np.ndarray:
    codes = np.ascontiguousarray(codes, dtype=np.int32)
    out = np.zeros(state_nbytes(codes.size, k), dtype=np.uint8)
    lib().fewbit_oracle_deflate(codes.ctypes.data, codes.size, out.ctypes.data, k)
    return out


def inflate(state, n: int, k: int) -> np.ndarray:
    state = np.ascontiguousarray(state, dtype=np.uint8)
    assert state.size >= state_nbytes(n, k)
    out = np.zeros(n, dtype=np.int32)
    lib().fewbit_oracle_inflate(out.ctypes.data, n, state.ctypes.data, k)
    return out


def searchsorted(x: torch.Tensor, borders: torch.Tensor) -> torch.Tensor:
    x, borders = _cpu(x), _cpu(borders).to(x.dtype)
    codes = torch.empty(x.numel(), dtype=torch.int32)
    lib().fewbit_oracle_searchsorted(_DT[x.dtype], x.data_ptr(), x.numel(), borders.data_ptr(), borders.numel(),
                                     codes.data_ptr())
    return codes.view(x.shape)


def activation(name: str, x: torch.Tensor, p0: float = 0.0, p1: float = 0.0) -> torch.Tensor:
    x = _cpu(x)
    y = torch.empty_like(x)
    lib().fewbit_oracle_activation(CONTINUOUS.index(name), _DT[x.dtype], x.data_ptr(), x.numel(), p0, p1,
                                   y.data_ptr())
    return y


def quantize(name: str, x: torch.Tensor, borders: torch.Tensor, p0: float = 0.0, p1: float = 0.0):
    """-> (y, state[k*ceil(n/8)] uint8, k); borders are the INNER borders, cast to x.dtype first."""
    x, borders = _cpu(x), _cpu(borders).to(x.dtype)
    k = bitwidth(borders.numel() + 1)
    y = torch.empty_like(x)
    state = torch.zeros(state_nbytes_padded(x.numel(), k), dtype=torch.uint8)
    lib().fewbit_oracle_quantize(CONTINUOUS.index(name), _DT[x.dtype], x.data_ptr(), x.numel(), borders.data_ptr(),
                                 borders.numel(), p0, p1, y.data_ptr(), state.data_ptr())
    return y, state, k


def quantize_backward(gy: torch.Tensor, state: torch.Tensor, levels: torch.Tensor) -> torch.Tensor:
    gy, state, levels = _cpu(gy), _cpu(state), _cpu(levels).to(gy.dtype)
    k = bitwidth(levels.numel())
    assert state.numel() >= state_nbytes(gy.numel(), k)
    gx = torch.empty_like(gy)
    lib().fewbit_oracle_quantize_backward(_DT[gy.dtype], gy.data_ptr(), gy.numel(), state.data_ptr(),
                                          levels.data_ptr(), levels.numel(), gx.data_ptr())
    return gx


def stepwise1_forward(name: str, x: torch.Tensor, p0: float = 0.0, p1: float = 0.0):
    x = _cpu(x)
    y = torch.empty_like(x)
    state = torch.zeros((x.numel() + 7) // 8, dtype=torch.uint8)
    lib().fewbit_oracle_stepwise1_forward(STEPWISE1.index(name), _DT[x.dtype], x.data_ptr(), x.numel(), p0, p1,
                                          y.data_ptr(), state.data_ptr())
    return y, state


def stepwise1_backward(name: str, gy: torch.Tensor, state: torch.Tensor, p0: float = 0.0) -> torch.Tensor:
    gy, state = _cpu(gy), _cpu(state)
    gx = torch.empty_like(gy)
    lib().fewbit_oracle_stepwise1_backward(STEPWISE1.index(name), _DT[gy.dtype], gy.data_ptr(), gy.numel(),
                                           state.data_ptr(), p0, gx.data_ptr())
    return gx


def convert(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Software dtype conversion of the C side (checked against torch in tests)."""
    x = _cpu(x)
    y = torch.empty(x.shape, dtype=dtype)
    lib().fewbit_oracle_convert(x.data_ptr(), _DT[x.dtype], y.data_ptr(), _DT[dtype], x.numel())
    return y


def ref_codec():
    """The reference's own codec.h compiled where it lies (oracle/_ref), or None."""
    so = HERE / '_ref' / 'libcodec_ref.so'
    if not so.exists():
        return None
    L = ctypes.CDLL(str(so))
    for f in (L.ref_deflate_u8, L.ref_inflate_u8):
        f.restype = None
        f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int32]
    return L
