"""Discrete cosine transforms (types 2 and 3) on torch tensors, with scipy's conventions.

Counterpart of the reference's ``fewbit/fft.py`` (``dct``/``idct``, :87-125), used by the ``'dct'`` sketch of
:func:`fewbit.functional.linear_grp`.  Written directly from the definitions

    DCT-II :  y_k = 2 * sum_n x_n cos(pi k (2n+1) / 2N)
    DCT-III:  y_k = x_0 + 2 * sum_{n>=1} x_n cos(pi n (2k+1) / 2N)

as one zero-padded 2N-point FFT each (rocFFT on the GPU): ``cos(pi k (2n+1)/2N) = Re(w^k e^{-2 pi i nk/2N})`` with
``w = e^{-i pi/2N}``.  Normalisation modes follow ``scipy.fft.dct``: ``'backward'`` (no scaling), ``'forward'``
(1/2N) and ``'ortho'`` (orthonormal matrix; type 3 is then the exact inverse and transpose of type 2).
"""
import math
from typing import Optional

import torch

__all__ = ('dct', 'idct')

_NORMS = ('backward', 'forward', 'ortho')


def _resize(x: torch.Tensor, n: Optional[int]) -> torch.Tensor:
    """Truncate or zero-pad the last dimension to ``n`` (scipy's ``n=`` argument)."""
    if n is None or n == x.shape[-1]:
        return x
    if n < 1:
        raise ValueError(f'Invalid number of data points ({n}) specified.')
    if n < x.shape[-1]:
        return x[..., :n]
    return torch.nn.functional.pad(x, (0, n - x.shape[-1]))


def _twiddle(n: int, sign: float, like: torch.Tensor) -> torch.Tensor:
    k = torch.arange(n, device=like.device, dtype=torch.float64)
    ang = sign * math.pi * k / (2 * n)
    return torch.complex(torch.cos(ang), torch.sin(ang)).to(torch.complex128 if like.dtype == torch.float64 else torch.complex64)


def _dct2_last(x: torch.Tensor, norm: str) -> torch.Tensor:
    n = x.shape[-1]
    spec = torch.fft.fft(x, n=2 * n, dim=-1)[..., :n]
    y = 2.0 * (spec * _twiddle(n, -1.0, x)).real
    if norm == 'forward':
        y = y / (2 * n)
    elif norm == 'ortho':
        y = y * math.sqrt(1.0 / (2 * n))
        y[..., 0] = y[..., 0] * math.sqrt(0.5)
    return y


def _dct3_last(x: torch.Tensor, norm: str) -> torch.Tensor:
    n = x.shape[-1]
    if norm == 'ortho':
        x = x * math.sqrt(1.0 / (2 * n))
        x = torch.cat([x[..., :1] * math.sqrt(2.0), x[..., 1:]], dim=-1)
    elif norm == 'forward':
        x = x / (2 * n)
    coef = x * _twiddle(n, 1.0, x) * 2.0
    coef = torch.cat([coef[..., :1] * 0.5, coef[..., 1:]], dim=-1)
    # sum_n coef_n e^{+2 pi i nk/2N}  =  2N * ifft(coef zero-padded to 2N)
    return (torch.fft.ifft(coef, n=2 * n, dim=-1)[..., :n] * (2 * n)).real


def _apply(x: torch.Tensor, kind: int, n: Optional[int], dim: int, norm: str) -> torch.Tensor:
    if norm not in _NORMS:
        raise ValueError(f'Unexpected normalization regime: {norm}.')
    if kind not in (2, 3):
        raise ValueError(f'Only DCT of types 2 and 3 are implemented, got type {kind}.')
    dtype = x.dtype
    work = x if dtype in (torch.float32, torch.float64) else x.float()       # FFTs of 16-bit inputs run in fp32
    work = _resize(work.movedim(dim, -1), n)
    out = _dct2_last(work, norm) if kind == 2 else _dct3_last(work, norm)
    return out.movedim(-1, dim).to(dtype)


def dct(x: torch.Tensor, type: int = 2, n: Optional[int] = None, dim: int = -1, norm: str = 'backward') -> torch.Tensor:
    """Discrete cosine transform of ``x`` along ``dim`` (``scipy.fft.dct`` semantics for types 2 and 3)."""
    return _apply(x, type, n, dim, norm)


def idct(x: torch.Tensor, type: int = 2, n: Optional[int] = None, dim: int = -1, norm: str = 'backward') -> torch.Tensor:
    """Inverse of :func:`dct` of the same ``type`` and ``norm`` (``scipy.fft.idct``)."""
    if norm not in _NORMS:
        raise ValueError(f'Unexpected normalization regime: {norm}.')
    inverse_norm = {'backward': 'forward', 'forward': 'backward', 'ortho': 'ortho'}[norm]
    return _apply(x, {2: 3, 3: 2}.get(type, type), n, dim, inverse_norm)
