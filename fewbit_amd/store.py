"""Table store: (function, bits) -> (borders[2^k+1], levels[2^k]) with per-(device, dtype) casts cached.

Mirrors the reference's ``StepwiseStore`` (fewbit/functional/activations.py:24-86): same methods, same npz
key scheme ``{func}{bits:02d}-borders`` / ``-levels``, same KeyError for an unknown table, and the same cast
``tensor.to(device, dtype)`` of the float64 tables (so the 16-bit borders are bit-identical to the reference's).
"""
from pathlib import Path
from typing import Dict, Iterator, Optional, Tuple, Union

import numpy as np
import torch

__all__ = ['StepwiseStore', 'store', 'BUILTIN_TABLES']

BUILTIN_TABLES = Path(__file__).resolve().parent / 'data' / 'builtin.npz'

Table = Tuple[torch.Tensor, torch.Tensor]


class StepwiseStore:
    """Stepwise approximations of activation-function derivatives, keyed by (name, bits)."""

    def __init__(self):
        self._store: Dict[Tuple[str, int], Table] = {}
        self._cache: Dict[Tuple[str, int, torch.device, torch.dtype], Table] = {}
        self._inner: Dict[Tuple[str, int, torch.device, torch.dtype], Table] = {}
        self.version = 0          # bumped by add(): lets callers that keep their own per-(device, dtype) casts notice

    def __len__(self) -> int:
        return len(self._store)

    def __repr__(self) -> str:
        return f'{type(self).__name__}(stored={len(self._store)}, cached={len(self._cache)})'

    def __contains__(self, key: Tuple[str, int]) -> bool:
        return tuple(key) in self._store

    def add(self, name: str, bits: int, value: Table) -> None:
        borders, values = value
        if borders.ndim != 1 or values.ndim != 1 or borders.numel() != values.numel() + 1:
            raise ValueError('Expected one-dimensional `borders` (with both sentinels) one longer than `levels`.')
        entry = (borders, values.to(borders))
        self._store[(name, int(bits))] = entry
        self.version += 1
        # drop stale casts of a replaced table
        for cache in (self._cache, self._inner):
            for key in [k for k in cache if k[:2] == (name, int(bits))]:
                del cache[key]
        self._cache[(name, int(bits), borders.device, borders.dtype)] = entry

    def get(self, name: str, bits: int, device: Union[None, str, torch.device] = None,
            dtype: Optional[torch.dtype] = None) -> Table:
        device = torch.device(device or 'cpu')
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        dtype = dtype or torch.float32
        key = (name, bits, device, dtype)
        hit = self._cache.get(key)
        if hit is not None:
            return hit
        leaf = self._store.get((name, bits))
        if leaf is None:
            raise KeyError(f'There is not {bits}-bit quantized gradients for activation function {name}.')
        # The cast outlives this call, so it must be an ordinary tensor even when the first call for this key happens under
        # torch.inference_mode(): a cached *inference* tensor could never be saved for backward by a later training call
        # ("Inference tensors cannot be saved for backward").
        with torch.inference_mode(False):
            cast = tuple(el.to(device, dtype, copy=True) if el.is_inference() else el.to(device, dtype) for el in leaf)
        self._cache[key] = cast
        return cast

    def get_inner(self, name: str, bits: int, device: torch.device, dtype: torch.dtype) -> Table:
        """``(borders[1:-1], levels)`` of :meth:`get`, contiguous and cached: what the kernels take, without a slice and
        two ``.to`` calls on every forward (the per-call overhead SURVEY 8a2 points at)."""
        key = (name, bits, device, dtype)
        hit = self._inner.get(key)
        if hit is None:
            borders, levels = self.get(name, bits, device, dtype)
            with torch.inference_mode(False):        # see get(): cached tensors must not be inference tensors
                hit = self._inner[key] = (borders[1:-1].contiguous(), levels)
        return hit

    def items(self, cached: bool = False) -> Iterator:
        yield from (self._cache if cached else self._store).items()

    def load(self, path: Union[str, Path]) -> 'StepwiseStore':
        """Add every table of an npz file whose arrays are named ``{func}{bits:02d}-borders|levels``."""
        with np.load(path) as npz:
            for key in sorted({k.split('-', 1)[0] for k in npz.keys()}):
                name, bits = key[:-2], int(key[-2:])
                self.add(name, bits, (torch.tensor(npz[f'{key}-borders']), torch.tensor(npz[f'{key}-levels'])))
        return self


store = StepwiseStore()
store.load(BUILTIN_TABLES)
