"""fewbit_amd -- MI355X (gfx950) implementation of FewBit's quantized-activation backward path.

``import fewbit`` (the thin alias package next to this one) gives the reference's surface:
``fewbit.functional.*``, ``fewbit.GELU(bits=3)`` & co, ``fewbit.map_module`` and ``torch.ops.fewbit.*``.

Native library: ``fewbit_amd/libfewbit.so`` (TORCH_LIBRARY(fewbit) glue) which links ``libfewbit_hip.so`` (HIP
kernels + C-ABI).  As in the reference (fewbit/__init__.py:17-23) the environment variable ``FEWBIT_NATIVE`` in
{0, no, false} skips loading it and a load failure is a ``RuntimeWarning``; unlike the reference there is no Python
fallback for GPU tensors afterwards -- they raise (host tensors keep working through plain PyTorch).
"""
from os import getenv
from pathlib import Path
from warnings import warn

import torch

# FEWBIT_OPS_LIB: another build of the SAME operator library (e.g. the public-API-only test build), not a fallback
_NATIVE_PATH = Path(getenv('FEWBIT_OPS_LIB') or Path(__file__).resolve().with_name('libfewbit.so'))
_native_error = 'not attempted'
_native_loaded = False


def native_loaded() -> bool:
    return _native_loaded


def native_error() -> str:
    return _native_error


def load_native(path=None) -> bool:
    """Load the operator library (idempotent).  Returns True when ``torch.ops.fewbit`` is backed by it."""
    global _native_loaded, _native_error
    if _native_loaded:
        return True
    try:
        torch.ops.load_library(str(path or _NATIVE_PATH))
        torch.ops.fewbit.gelu  # noqa: B018  (AttributeError/RuntimeError if the registration is missing)
        _native_loaded, _native_error = True, ''
    except Exception as e:  # noqa: BLE001  -- OSError, RuntimeError, AttributeError all mean "not available"
        _native_error = f'{type(e).__name__}: {e}'
    return _native_loaded


def autograd_route(name: str, on=None) -> bool:
    """Query (``on=None``) or set one of the operator library's autograd routes: ``'direct_node'`` (hand-written backward
    node instead of a ``torch::autograd::Function``), ``'base_dirty'`` (in place on a whole-tensor view modifies the base)
    and ``'fresh_view'`` (that call returns a new view of the base).  Returns the previous setting.  All routes give the
    same values, gradients and saved bytes (fewbit_amd/csrc/torch_ops.cpp, header comment); the switches exist so that the
    public-API fallback stays tested.  Environment: ``FEWBIT_NO_DIRECT_NODE`` / ``FEWBIT_NO_BASE_DIRTY`` /
    ``FEWBIT_NO_FRESH_VIEW`` = 1."""
    import ctypes
    if not _native_loaded:
        raise RuntimeError(f'operator library not loaded: {_native_error}')
    lib = ctypes.CDLL(str(_NATIVE_PATH))
    lib.fewbit_torch_route.argtypes, lib.fewbit_torch_route.restype = [ctypes.c_char_p, ctypes.c_int], ctypes.c_int
    prev = lib.fewbit_torch_route(name.encode(), -1 if on is None else int(bool(on)))
    if prev == -1:
        raise KeyError(name)
    if prev == -2:
        raise RuntimeError(f'route {name!r} needs a library built with FEWBIT_AUTOGRAD_INTERNALS=1')
    return bool(prev)


def autograd_internals() -> bool:
    """Was the operator library built with the internal-API autograd routes (only for the torch release it was verified on)?"""
    import ctypes
    lib = ctypes.CDLL(str(_NATIVE_PATH))
    lib.fewbit_torch_autograd_internals.restype = ctypes.c_int
    return bool(lib.fewbit_torch_autograd_internals())


if getenv('FEWBIT_NATIVE') not in ('0', 'no', 'false'):
    if not load_native():
        warn(f'Failed to load ops library: {_native_error}.', RuntimeWarning)
else:
    _native_error = 'disabled by FEWBIT_NATIVE'

from . import functional  # noqa: E402,F401
from . import modules  # noqa: E402,F401
from . import approx, cli, compat, fft, linear, variance  # noqa: E402,F401
from .modules import *  # noqa: E402,F401,F403
from .linear import LinearCRS, LinearGRP, RandomizedLinear  # noqa: E402,F401
from .util import map_module, memory_usage_hooks  # noqa: E402,F401

__version__ = '0.2.0'
