"""ctypes binding of the C-ABI library ``libfewbit_hip.so`` (include/fewbit_hip.h).

This is the thin, torch-free-in-signature boundary the parity tests and ``bench.py`` drive directly:
device pointers, sizes and a ``hipStream_t``.  There is no fallback: if the shared library is missing
every call raises (the product never computes this path on the CPU).
"""
import ctypes
from pathlib import Path
from typing import Optional

import torch

__all__ = [
    'CONTINUOUS', 'STEPWISE1', 'LIB_PATH', 'lib', 'loaded', 'bitwidth', 'state_nbytes', 'quantize_forward',
    'quantize_backward', 'bind_forward', 'bind_backward', 'bind_stepwise1_forward', 'bind_stepwise1_backward',
    'stepwise1_forward', 'stepwise1_backward', 'pack_codes', 'unpack_codes', 'FewbitHipError', 'describe_forward',
    'describe_backward', 'describe_stepwise1_forward', 'describe_stepwise1_backward', 'tune',
    'SKETCH_DISTS', 'ABI_VERSION', 'sketch', 'next_sketch_seed', 'mix_sketch_seed', 'sketch_matrix', 'sketch_workspace_bytes', 'describe_sketch', 'tune_sketch_slices', 'tune_sketch_waves', 'tune_sketch_halves', 'tune_sketch_convert', 'tune_sketch_partials', 'tune_sketch_materialise', 'xoshiro128pp',
    'philox4x32', 'sampled_dct', 'sampled_dct_seeded', 'sampled_rows', 'sampled_dct_workspace_bytes',
]

import os

# FEWBIT_HIP_LIB: kernel-tuning hook (an alternative build of the same library), not a fallback
LIB_PATH = Path(os.environ.get('FEWBIT_HIP_LIB') or Path(__file__).resolve().with_name('libfewbit_hip.so'))

# enum order of include/fewbit_hip.h
CONTINUOUS = ('celu', 'elu', 'gelu', 'hardswish', 'logsigmoid', 'mish', 'selu', 'sigmoid', 'silu', 'softplus',
              'softsign', 'tanh', 'tanhshrink', 'identity', 'identity_fold')
STEPWISE1 = ('hardshrink', 'hardsigmoid', 'hardtanh', 'leaky_relu', 'relu', 'relu6', 'softshrink', 'threshold')
DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
SKETCH_DISTS = ('rademacher', 'gaussian')      # enum fewbit_sketch_dist
ABI_VERSION = 5                                # FEWBIT_HIP_ABI_VERSION this binding was written against

# every symbol include/fewbit_hip.h declares
SYMBOLS = ('fewbit_hip_abi_version', 'fewbit_hip_last_error', 'fewbit_hip_bitwidth', 'fewbit_hip_state_nbytes',
           'fewbit_hip_quantize_forward', 'fewbit_hip_quantize_backward', 'fewbit_hip_stepwise1_forward',
           'fewbit_hip_stepwise1_backward', 'fewbit_hip_pack_codes', 'fewbit_hip_unpack_codes',
           'fewbit_hip_describe_quantize_forward', 'fewbit_hip_describe_quantize_backward',
           'fewbit_hip_describe_stepwise1_forward', 'fewbit_hip_describe_stepwise1_backward', 'fewbit_hip_tune',
           'fewbit_hip_sketch_workspace', 'fewbit_hip_sketch', 'fewbit_hip_sketch_device_seed', 'fewbit_hip_sketch_next_seed',
           'fewbit_hip_sketch_mix_seed', 'fewbit_hip_sketch_matrix', 'fewbit_hip_sketch_describe',
           'fewbit_hip_sampled_dct_workspace', 'fewbit_hip_sampled_dct', 'fewbit_hip_sampled_dct_seeded', 'fewbit_hip_sampled_rows',
           'fewbit_hip_philox4x32', 'fewbit_hip_xoshiro128pp')


class FewbitHipError(RuntimeError):
    pass


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise FewbitHipError(f'{LIB_PATH} is missing: build it with `make -C fewbit_amd/csrc` '
                                 '(or `python -c "import __graft_entry__ as g; g.build()"`)')
        L = ctypes.CDLL(str(LIB_PATH))
        vp, sz, i32, dbl = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double
        L.fewbit_hip_abi_version.restype = i32
        L.fewbit_hip_abi_version.argtypes = []
        have = L.fewbit_hip_abi_version()
        if have < ABI_VERSION:       # (an older build handed in through FEWBIT_HIP_LIB: say so instead of an AttributeError)
            raise FewbitHipError(f'{LIB_PATH} has ABI version {have}, this binding needs {ABI_VERSION}: rebuild it '
                                 '(make -C fewbit_amd/csrc)')
        L.fewbit_hip_last_error.restype = ctypes.c_char_p
        L.fewbit_hip_last_error.argtypes = []
        L.fewbit_hip_bitwidth.restype = i32
        L.fewbit_hip_bitwidth.argtypes = [i32]
        L.fewbit_hip_state_nbytes.restype = sz
        L.fewbit_hip_state_nbytes.argtypes = [sz, i32]
        L.fewbit_hip_quantize_forward.restype = i32
        L.fewbit_hip_quantize_forward.argtypes = [i32, i32, vp, vp, vp, sz, vp, i32, dbl, dbl, vp]
        L.fewbit_hip_quantize_backward.restype = i32
        L.fewbit_hip_quantize_backward.argtypes = [i32, vp, vp, vp, sz, vp, i32, vp]
        L.fewbit_hip_stepwise1_forward.restype = i32
        L.fewbit_hip_stepwise1_forward.argtypes = [i32, i32, vp, vp, vp, sz, dbl, dbl, vp]
        L.fewbit_hip_stepwise1_backward.restype = i32
        L.fewbit_hip_stepwise1_backward.argtypes = [i32, i32, vp, vp, vp, sz, dbl, vp]
        L.fewbit_hip_pack_codes.restype = i32
        L.fewbit_hip_pack_codes.argtypes = [vp, vp, sz, i32, vp]
        L.fewbit_hip_unpack_codes.restype = i32
        L.fewbit_hip_unpack_codes.argtypes = [vp, vp, sz, i32, vp]
        cp = ctypes.c_char_p
        L.fewbit_hip_describe_quantize_forward.restype = i32
        L.fewbit_hip_describe_quantize_forward.argtypes = [i32, i32, sz, i32, cp, sz]
        L.fewbit_hip_describe_quantize_backward.restype = i32
        L.fewbit_hip_describe_quantize_backward.argtypes = [i32, sz, i32, cp, sz]
        L.fewbit_hip_describe_stepwise1_forward.restype = i32
        L.fewbit_hip_describe_stepwise1_forward.argtypes = [i32, i32, sz, cp, sz]
        L.fewbit_hip_describe_stepwise1_backward.restype = i32
        L.fewbit_hip_describe_stepwise1_backward.argtypes = [i32, i32, sz, cp, sz]
        L.fewbit_hip_tune.restype = i32
        L.fewbit_hip_tune.argtypes = [cp, ctypes.c_longlong]
        u64 = ctypes.c_uint64
        L.fewbit_hip_sketch_workspace.restype = sz
        L.fewbit_hip_sketch_workspace.argtypes = [i32, i32, sz, sz, sz]
        L.fewbit_hip_sketch.restype = i32
        L.fewbit_hip_sketch.argtypes = [i32, i32, vp, sz, sz, sz, sz, u64, dbl, vp, vp, sz, vp]
        L.fewbit_hip_sketch_device_seed.restype = i32
        L.fewbit_hip_sketch_device_seed.argtypes = [i32, i32, vp, sz, sz, sz, sz, vp, dbl, vp, vp, sz, vp]
        L.fewbit_hip_sketch_next_seed.restype = i32
        L.fewbit_hip_sketch_next_seed.argtypes = [vp, u64, vp, vp]
        L.fewbit_hip_sketch_mix_seed.restype = u64
        L.fewbit_hip_sketch_mix_seed.argtypes = [u64, u64]
        L.fewbit_hip_sketch_matrix.restype = i32
        L.fewbit_hip_sketch_matrix.argtypes = [i32, i32, u64, sz, sz, sz, sz, vp, vp]
        L.fewbit_hip_sketch_describe.restype = i32
        L.fewbit_hip_sketch_describe.argtypes = [i32, i32, sz, sz, sz, cp, sz]
        L.fewbit_hip_sampled_dct_workspace.restype = sz
        L.fewbit_hip_sampled_dct_workspace.argtypes = [i32, sz, sz, sz]
        L.fewbit_hip_sampled_dct.restype = i32
        L.fewbit_hip_sampled_dct.argtypes = [i32, vp, sz, sz, sz, vp, sz, dbl, vp, vp, sz, vp]
        L.fewbit_hip_sampled_dct_seeded.restype = i32
        L.fewbit_hip_sampled_dct_seeded.argtypes = [i32, vp, sz, sz, sz, ctypes.c_uint64, vp, sz, dbl, vp, vp, sz, vp]
        L.fewbit_hip_sampled_rows.restype = i32
        L.fewbit_hip_sampled_rows.argtypes = [ctypes.c_uint64, sz, sz, vp]
        L.fewbit_hip_philox4x32.restype = None
        L.fewbit_hip_philox4x32.argtypes = [ctypes.POINTER(ctypes.c_uint32)] * 3
        L.fewbit_hip_xoshiro128pp.restype = None
        L.fewbit_hip_xoshiro128pp.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), sz]
        # FEWBIT_SKETCH_MATERIALISE=0|1: the Gaussian sketch never through the S-from-memory path / also on the narrow layers of an fp32
        # model (the built-in policy takes it for 16-bit input, and for fp32 input on layers of 2048 features or more; the 1 GiB cap
        # on the fragments and the one-tile rule hold under either setting)
        forced = os.environ.get('FEWBIT_SKETCH_MATERIALISE', '')
        if forced in ('0', '1'):
            L.fewbit_hip_tune(b'sketch_materialise', int(forced))
        _lib = L
    return _lib


def loaded() -> bool:
    return _lib is not None


def _check(rc: int):
    if rc != 0:
        raise FewbitHipError(f'fewbit_hip error {rc}: {lib().fewbit_hip_last_error().decode()}')


def _dev(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.device.type != 'cuda':
        raise FewbitHipError(f'{what} must live on the GPU (got {t.device})')
    if not t.is_contiguous():
        raise FewbitHipError(f'{what} must be contiguous')
    return t


def _stream(stream: Optional[int], device: Optional[torch.device] = None) -> int:
    """``stream`` or torch's current stream ON THE TENSORS' DEVICE (not on whatever device happens to be current)."""
    return torch.cuda.current_stream(device).cuda_stream if stream is None else stream


class _on:
    """Make the tensors' device current around a launch: the library launches on the calling thread's current device
    (include/fewbit_hip.h), so a tensor on cuda:3 must not be launched while cuda:0 is current.  A no-op (no runtime
    call at all) in the common case that the device is current already."""
    __slots__ = ('index', 'prev')

    def __init__(self, device: torch.device):
        self.index = device.index if device.index is not None else torch.cuda.current_device()

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.prev != self.index:
            torch.cuda.set_device(self.index)
        return self

    def __exit__(self, *exc):
        if self.prev != self.index:
            torch.cuda.set_device(self.prev)
        return False


def _same_device(first: torch.Tensor, *others: torch.Tensor):
    for t in others:
        if t.device != first.device:
            raise FewbitHipError(f'tensors live on different devices ({first.device} and {t.device})')


def bitwidth(nlevels: int) -> int:
    return lib().fewbit_hip_bitwidth(nlevels)


def state_nbytes(n: int, nbits: int) -> int:
    return lib().fewbit_hip_state_nbytes(n, nbits)


def quantize_forward(fn: str, x: torch.Tensor, borders: torch.Tensor, p0: float = 0.0, p1: float = 0.0,
                     out: Optional[torch.Tensor] = None, state: Optional[torch.Tensor] = None,
                     stream: Optional[int] = None):
    """-> (y, state).  ``out=x`` runs in place like the reference op; ``borders`` are the inner borders."""
    x, borders = _dev(x, 'x'), _dev(borders, 'borders')
    if borders.dtype != x.dtype:
        raise FewbitHipError(f'borders dtype {borders.dtype} != input dtype {x.dtype}')
    k = bitwidth(borders.numel() + 1)
    with _on(x.device):
        y = torch.empty_like(x) if out is None else _dev(out, 'out')
        if state is None:
            state = torch.empty(state_nbytes(x.numel(), k), dtype=torch.uint8, device=x.device)
        _same_device(x, borders, y, state)
        _check(lib().fewbit_hip_quantize_forward(CONTINUOUS.index(fn), DTYPES[x.dtype], x.data_ptr(), y.data_ptr(),
                                                 state.data_ptr(), x.numel(), borders.data_ptr(), borders.numel(),
                                                 p0, p1, _stream(stream, x.device)))
    return y, state


def quantize_backward(gy: torch.Tensor, state: torch.Tensor, levels: torch.Tensor,
                      out: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
    gy, state, levels = _dev(gy, 'gy'), _dev(state, 'state'), _dev(levels, 'levels')
    if levels.dtype != gy.dtype:
        raise FewbitHipError(f'levels dtype {levels.dtype} != grad dtype {gy.dtype}')
    k = bitwidth(levels.numel())
    if state.numel() < state_nbytes(gy.numel(), k):
        raise FewbitHipError('state buffer too small')
    with _on(gy.device):
        gx = torch.empty_like(gy) if out is None else _dev(out, 'out')
        _same_device(gy, state, levels, gx)
        _check(lib().fewbit_hip_quantize_backward(DTYPES[gy.dtype], gy.data_ptr(), state.data_ptr(), gx.data_ptr(),
                                                  gy.numel(), levels.data_ptr(), levels.numel(),
                                                  _stream(stream, gy.device)))
    return gx


def _bound(f, args, device: torch.device, keepalive):
    """zero-argument launch of ``f(*args)`` with ``device`` current (see _on); one integer compare when it already is"""
    index = device.index if device.index is not None else torch.cuda.current_device()
    current, set_device = torch.cuda.current_device, torch.cuda.set_device

    def launch():
        prev = current()
        if prev != index:
            set_device(index)
            try:
                rc = f(*args)
            finally:
                set_device(prev)
        else:
            rc = f(*args)
        if rc:
            _check(rc)
    launch.keepalive = keepalive
    return launch


def bind_forward(fn: str, x: torch.Tensor, borders: torch.Tensor, out: torch.Tensor, state: torch.Tensor,
                 p0: float = 0.0, p1: float = 0.0, stream: Optional[int] = None):
    """Pre-resolved launch: returns a zero-argument callable that enqueues exactly this forward (same pointers,
    same stream, on the tensors' device) -- for loops where the per-call Python argument handling would otherwise
    out-weigh a ~10 us kernel."""
    x, borders, out, state = _dev(x, 'x'), _dev(borders, 'borders'), _dev(out, 'out'), _dev(state, 'state')
    if borders.dtype != x.dtype or out.dtype != x.dtype:
        raise FewbitHipError('x, borders and out must share one dtype')
    _same_device(x, borders, out, state)
    k = bitwidth(borders.numel() + 1)
    if state.numel() < state_nbytes(x.numel(), k) or out.numel() != x.numel():
        raise FewbitHipError('out/state buffers do not match the input size')
    args = (CONTINUOUS.index(fn), DTYPES[x.dtype], x.data_ptr(), out.data_ptr(), state.data_ptr(), x.numel(),
            borders.data_ptr(), borders.numel(), p0, p1, _stream(stream, x.device))
    return _bound(lib().fewbit_hip_quantize_forward, args, x.device, (x, borders, out, state))


def bind_backward(gy: torch.Tensor, state: torch.Tensor, levels: torch.Tensor, out: torch.Tensor,
                  stream: Optional[int] = None):
    """Pre-resolved launch of the backward, see bind_forward."""
    gy, state, levels, out = _dev(gy, 'gy'), _dev(state, 'state'), _dev(levels, 'levels'), _dev(out, 'out')
    if levels.dtype != gy.dtype or out.dtype != gy.dtype:
        raise FewbitHipError('gy, levels and out must share one dtype')
    _same_device(gy, state, levels, out)
    k = bitwidth(levels.numel())
    if state.numel() < state_nbytes(gy.numel(), k) or out.numel() != gy.numel():
        raise FewbitHipError('out/state buffers do not match the input size')
    args = (DTYPES[gy.dtype], gy.data_ptr(), state.data_ptr(), out.data_ptr(), gy.numel(), levels.data_ptr(),
            levels.numel(), _stream(stream, gy.device))
    return _bound(lib().fewbit_hip_quantize_backward, args, gy.device, (gy, state, levels, out))


def bind_stepwise1_forward(fn: str, x: torch.Tensor, out: torch.Tensor, state: torch.Tensor, p0: float = 0.0,
                           p1: float = 0.0, stream: Optional[int] = None):
    """Pre-resolved launch of a 1-bit forward, see bind_forward."""
    x, out, state = _dev(x, 'x'), _dev(out, 'out'), _dev(state, 'state')
    if out.dtype != x.dtype or out.numel() != x.numel() or state.numel() < state_nbytes(x.numel(), 1):
        raise FewbitHipError('out/state buffers do not match the input')
    _same_device(x, out, state)
    args = (STEPWISE1.index(fn), DTYPES[x.dtype], x.data_ptr(), out.data_ptr(), state.data_ptr(), x.numel(), p0, p1,
            _stream(stream, x.device))
    return _bound(lib().fewbit_hip_stepwise1_forward, args, x.device, (x, out, state))


def bind_stepwise1_backward(fn: str, gy: torch.Tensor, state: torch.Tensor, out: torch.Tensor, p0: float = 0.0,
                            stream: Optional[int] = None):
    """Pre-resolved launch of a 1-bit backward, see bind_forward."""
    gy, state, out = _dev(gy, 'gy'), _dev(state, 'state'), _dev(out, 'out')
    if out.dtype != gy.dtype or out.numel() != gy.numel() or state.numel() < state_nbytes(gy.numel(), 1):
        raise FewbitHipError('out/state buffers do not match the gradient')
    _same_device(gy, state, out)
    args = (STEPWISE1.index(fn), DTYPES[gy.dtype], gy.data_ptr(), state.data_ptr(), out.data_ptr(), gy.numel(), p0,
            _stream(stream, gy.device))
    return _bound(lib().fewbit_hip_stepwise1_backward, args, gy.device, (gy, state, out))


def stepwise1_forward(fn: str, x: torch.Tensor, p0: float = 0.0, p1: float = 0.0,
                      out: Optional[torch.Tensor] = None, state: Optional[torch.Tensor] = None,
                      stream: Optional[int] = None):
    x = _dev(x, 'x')
    with _on(x.device):
        y = torch.empty_like(x) if out is None else _dev(out, 'out')
        if state is None:
            state = torch.empty(state_nbytes(x.numel(), 1), dtype=torch.uint8, device=x.device)
        _same_device(x, y, state)
        _check(lib().fewbit_hip_stepwise1_forward(STEPWISE1.index(fn), DTYPES[x.dtype], x.data_ptr(), y.data_ptr(),
                                                  state.data_ptr(), x.numel(), p0, p1, _stream(stream, x.device)))
    return y, state


def stepwise1_backward(fn: str, gy: torch.Tensor, state: torch.Tensor, p0: float = 0.0,
                       out: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
    gy, state = _dev(gy, 'gy'), _dev(state, 'state')
    with _on(gy.device):
        gx = torch.empty_like(gy) if out is None else _dev(out, 'out')
        _same_device(gy, state, gx)
        _check(lib().fewbit_hip_stepwise1_backward(STEPWISE1.index(fn), DTYPES[gy.dtype], gy.data_ptr(),
                                                   state.data_ptr(), gx.data_ptr(), gy.numel(), p0,
                                                   _stream(stream, gy.device)))
    return gx


def pack_codes(codes: torch.Tensor, nbits: int, stream: Optional[int] = None) -> torch.Tensor:
    codes = _dev(codes, 'codes')
    assert codes.dtype == torch.int32
    with _on(codes.device):
        state = torch.empty(state_nbytes(codes.numel(), nbits), dtype=torch.uint8, device=codes.device)
        _check(lib().fewbit_hip_pack_codes(codes.data_ptr(), state.data_ptr(), codes.numel(), nbits,
                                           _stream(stream, codes.device)))
    return state


def unpack_codes(state: torch.Tensor, n: int, nbits: int, stream: Optional[int] = None) -> torch.Tensor:
    state = _dev(state, 'state')
    with _on(state.device):
        codes = torch.empty(n, dtype=torch.int32, device=state.device)
        _check(lib().fewbit_hip_unpack_codes(state.data_ptr(), codes.data_ptr(), n, nbits, _stream(stream, state.device)))
    return codes


# ---- what a call would launch (kernel instantiation + launch shape), and run-time launch tuning -------------------
def _describe(call, *args, device=None) -> dict:
    import json
    buf = ctypes.create_string_buffer(512)
    with _on(torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)):
        _check(call(*args, buf, len(buf)))
    return json.loads(buf.value.decode())


def describe_forward(fn: str, dtype: torch.dtype, n: int, nborders: int, device=None) -> dict:
    """Kernel and launch shape ``quantize_forward`` would use for ``n`` elements (nothing is launched)."""
    return _describe(lib().fewbit_hip_describe_quantize_forward, CONTINUOUS.index(fn), DTYPES[dtype], n, nborders, device=device)


def describe_backward(dtype: torch.dtype, n: int, nlevels: int, device=None) -> dict:
    return _describe(lib().fewbit_hip_describe_quantize_backward, DTYPES[dtype], n, nlevels, device=device)


def describe_stepwise1_forward(fn: str, dtype: torch.dtype, n: int, device=None) -> dict:
    return _describe(lib().fewbit_hip_describe_stepwise1_forward, STEPWISE1.index(fn), DTYPES[dtype], n, device=device)


def describe_stepwise1_backward(fn: str, dtype: torch.dtype, n: int, device=None) -> dict:
    return _describe(lib().fewbit_hip_describe_stepwise1_backward, STEPWISE1.index(fn), DTYPES[dtype], n, device=device)


def tune(**settings: int) -> None:
    """Set launch-tuning keys of the library (see fewbit_hip_tune in include/fewbit_hip.h); -1 = built-in policy."""
    for key, value in settings.items():
        _check(lib().fewbit_hip_tune(key.encode(), int(value)))


# ---- random-projection products (fewbit_amd/csrc/fewbit_sketch.hip): out = scale * S . m, S a function of the seed ------
def sketch_workspace_bytes(dist: str, rows: int, features: int, proj: int, dtype: torch.dtype = torch.bfloat16) -> int:
    return lib().fewbit_hip_sketch_workspace(SKETCH_DISTS.index(dist), DTYPES[dtype], rows, features, proj)


def sketch(dist: str, m: torch.Tensor, proj: int, seed, scale: float = 1.0, out: Optional[torch.Tensor] = None,
           workspace: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
    """``scale * S @ m`` for the ``proj x rows`` random matrix S(seed) of kind ``dist`` ('rademacher' / 'gaussian'), which is
    never kept (generated inside the product kernel, or -- Gaussian, 16-bit ``m`` wider than 256 features -- written once into the
    workspace as MFMA fragments and read back).  ``m``: 2-D, rows x features, unit stride along the features (any row stride).
    ``workspace``: ``sketch_workspace_bytes(...)`` bytes of scratch, allocated here when not given.  ``seed``: an int, or a
    one-element int64 tensor on the device of ``m`` whose value is read when the kernel runs (``next_sketch_seed``)."""
    if m.device.type != 'cuda':
        raise FewbitHipError(f'm must live on the GPU (got {m.device})')
    if m.dim() != 2 or (m.shape[1] > 1 and m.stride(1) != 1):
        raise FewbitHipError('m must be 2-D with unit stride along its last dimension')
    if m.dtype not in DTYPES:
        raise FewbitHipError(f'unsupported dtype {m.dtype}')
    rows, features = m.shape
    ld = m.stride(0) if rows > 1 else features
    with _on(m.device):
        if out is None:
            out = torch.empty((proj, features), dtype=m.dtype, device=m.device)
        elif out.shape != (proj, features) or out.dtype != m.dtype or not out.is_contiguous():
            raise FewbitHipError('out must be a contiguous proj x features tensor of the dtype of m')
        need = sketch_workspace_bytes(dist, rows, features, proj, m.dtype)
        if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
            workspace = torch.empty(need, dtype=torch.uint8, device=m.device)
        _same_device(m, out, *(() if workspace is None else (workspace, )))
        tail = (scale, out.data_ptr(), 0 if workspace is None else workspace.data_ptr(),
                0 if workspace is None else workspace.numel() * workspace.element_size(), _stream(stream, m.device))
        if isinstance(seed, torch.Tensor):
            _seed_word(seed, 'seed')
            _same_device(m, seed)
            _check(lib().fewbit_hip_sketch_device_seed(SKETCH_DISTS.index(dist), DTYPES[m.dtype], m.data_ptr(), rows, features, ld,
                                                       proj, seed.data_ptr(), *tail))
        else:
            _check(lib().fewbit_hip_sketch(SKETCH_DISTS.index(dist), DTYPES[m.dtype], m.data_ptr(), rows, features, ld, proj,
                                           seed & 0xffffffffffffffff, *tail))
    return out


def _seed_word(t: torch.Tensor, what: str) -> None:
    if t.device.type != 'cuda' or t.dtype != torch.int64 or t.numel() != 1:
        raise FewbitHipError(f'{what} must be a one-element int64 tensor on the GPU')


def next_sketch_seed(counter: torch.Tensor, base: int, out: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
    """On the device, in stream order: ``out = mix_sketch_seed(base, counter); counter += 1``.  Recorded into a hipGraph this
    gives every replay (and every call inside it) its own seed; pass ``out`` to ``sketch`` as the seed."""
    _seed_word(counter, 'counter')
    with _on(counter.device):
        if out is None:
            out = torch.empty(1, dtype=torch.int64, device=counter.device)
        _seed_word(out, 'out')
        _same_device(counter, out)
        _check(lib().fewbit_hip_sketch_next_seed(counter.data_ptr(), base & 0xffffffffffffffff, out.data_ptr(), _stream(stream, counter.device)))
    return out


def mix_sketch_seed(base: int, count: int) -> int:
    """host evaluation of the seed ``next_sketch_seed`` leaves for counter value ``count``"""
    return int(lib().fewbit_hip_sketch_mix_seed(base & 0xffffffffffffffff, count & 0xffffffffffffffff))


def sketch_matrix(dist: str, dtype: torch.dtype, seed: int, nrows: int, ncols: int, row0: int = 0, col0: int = 0,
                  device='cuda', stream: Optional[int] = None) -> torch.Tensor:
    """S[row0:row0+nrows, col0:col0+ncols] as fp32, rounded as the product kernel rounds its operand for ``dtype`` (test seam)."""
    device = torch.device(device)
    with _on(device if device.index is not None else torch.device('cuda', torch.cuda.current_device())):
        out = torch.empty((nrows, ncols), dtype=torch.float32, device=device)
        _check(lib().fewbit_hip_sketch_matrix(SKETCH_DISTS.index(dist), DTYPES[dtype], seed & 0xffffffffffffffff, row0, col0,
                                              nrows, ncols, out.data_ptr(), _stream(stream, out.device)))
    return out


def describe_sketch(dist: str, rows: int, features: int, proj: int, dtype: torch.dtype = torch.bfloat16, device=None) -> dict:
    return _describe(lib().fewbit_hip_sketch_describe, SKETCH_DISTS.index(dist), DTYPES[dtype], rows, features, proj, device=device)


# test and measurement seams: the random-projection keys of the one tuning hook (fewbit_hip_tune), -1 = built-in policy
def tune_sketch_slices(slices: int) -> None:
    """force the number of row slices"""
    tune(sketch_slices=slices)


def tune_sketch_waves(waves: int) -> None:
    """waves per workgroup: 4 (128-row tile) or 8 (256-row tile)"""
    tune(sketch_waves=waves)


def tune_sketch_halves(halves: int) -> None:
    """column halves per workgroup: 1 or 2 (the 128 x 512 tile)"""
    tune(sketch_halves=halves)


def tune_sketch_convert(convert: int) -> None:
    """round fp32 input to bf16 in one pass before the product (1), never (0)"""
    tune(sketch_convert=convert)


def tune_sketch_partials(bf16_partials: int) -> None:
    """partial sums of sliced bf16 products in bf16 (1 / -1, the policy; 2: bf16 results only) or always in fp32 (0)"""
    tune(sketch_partials=bf16_partials)


def tune_sketch_materialise(materialise: int) -> None:
    """Gaussian S written to the workspace once as MFMA fragments and read back by the product kernel also on narrow fp32 layers (1),
    always regenerated inside the product kernel (0)"""
    tune(sketch_materialise=materialise)


# ---- sampled cosine transform (fewbit_amd/csrc/fewbit_dct.hip): out = scale * dct(m, dim=0, norm='ortho')[idx] -------------------
def sampled_dct_workspace_bytes(rows: int, features: int, proj: int, dtype: torch.dtype = torch.bfloat16) -> int:
    """bytes of scratch a ``sampled_dct`` call needs; 0 = this shape has no kernel (rows none of 2^k in [256, 262144], 3 x 2^k in [768, 49152], 5 x 2^k in [1280, 40960])"""
    if dtype not in DTYPES:
        return 0
    return lib().fewbit_hip_sampled_dct_workspace(DTYPES[dtype], rows, features, proj)


def _sampled_dct_call(m: torch.Tensor, proj: int, out: Optional[torch.Tensor], workspace: Optional[torch.Tensor], others, launch) -> torch.Tensor:
    if m.device.type != 'cuda':
        raise FewbitHipError(f'm must live on the GPU (got {m.device})')
    if m.dim() != 2 or (m.shape[1] > 1 and m.stride(1) != 1):
        raise FewbitHipError('m must be 2-D with unit stride along its last dimension')
    if m.dtype not in DTYPES:
        raise FewbitHipError(f'unsupported dtype {m.dtype}')
    rows, features = m.shape
    ld = m.stride(0) if rows > 1 else features
    need = sampled_dct_workspace_bytes(rows, features, proj, m.dtype)
    if need == 0 and proj and features:
        raise FewbitHipError(f'sampled_dct: no kernel for {rows} rows (2^k in [256, 262144], 3 x 2^k in [768, 49152] or 5 x 2^k in [1280, 40960] is needed)')
    with _on(m.device):
        if out is None:
            out = torch.empty((proj, features), dtype=m.dtype, device=m.device)
        elif out.shape != (proj, features) or out.dtype != m.dtype or not out.is_contiguous():
            raise FewbitHipError('out must be a contiguous proj x features tensor of the dtype of m')
        if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
            workspace = torch.empty(need, dtype=torch.uint8, device=m.device)
        _same_device(m, out, *others, *(() if workspace is None else (workspace, )))
        _check(launch(DTYPES[m.dtype], m.data_ptr(), rows, features, ld, out.data_ptr(), 0 if workspace is None else workspace.data_ptr(),
                      0 if workspace is None else workspace.numel() * workspace.element_size()))
    return out


def sampled_dct(m: torch.Tensor, idx: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None,
                workspace: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
    """``scale * dct(m, dim=0, norm='ortho')[idx]`` (DCT-II along the rows, orthonormal; the reference's 'dct' sketch) for a 2-D
    ``m`` (rows x features, unit stride along the features) whose row count is 2^k in [256, 262144], 3 x 2^k in [768, 49152] or 5 x 2^k in [1280, 40960]; ``idx``: int64 row
    numbers on the device of ``m``.  fp32 arithmetic, result in the dtype of ``m``."""
    if idx.dtype != torch.int64 or idx.dim() != 1 or idx.device != m.device or not idx.is_contiguous():
        raise FewbitHipError('idx must be a contiguous 1-D int64 tensor on the device of m')
    proj = idx.numel()
    return _sampled_dct_call(m, proj, out, workspace, (idx, ), lambda dt, mp, rows, features, ld, op, wp, wb: lib().fewbit_hip_sampled_dct(
        dt, mp, rows, features, ld, idx.data_ptr(), proj, scale, op, wp, wb, _stream(stream, m.device)))


def sampled_dct_seeded(m: torch.Tensor, proj: int, seed, scale: float = 1.0, out: Optional[torch.Tensor] = None,
                       workspace: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
    """``sampled_dct(m, sampled_rows(seed, rows, proj))`` without the array of row numbers: the rows are a function of ``seed`` that the
    kernel evaluates itself.  ``seed``: an int, or a one-element int64 tensor on the device of ``m`` whose value is read when the kernel
    runs (``next_sketch_seed``: a launch recorded into a hipGraph then samples fresh rows on every replay)."""
    if isinstance(seed, torch.Tensor):
        _seed_word(seed, 'seed')
        value, word, others = 0, seed.data_ptr(), (seed, )
    else:
        value, word, others = seed & 0xffffffffffffffff, 0, ()
    return _sampled_dct_call(m, proj, out, workspace, others, lambda dt, mp, rows, features, ld, op, wp, wb: lib().fewbit_hip_sampled_dct_seeded(
        dt, mp, rows, features, ld, value, word, proj, scale, op, wp, wb, _stream(stream, m.device)))


def sampled_rows(seed: int, rows: int, proj: int) -> torch.Tensor:
    """The row numbers ``sampled_dct_seeded`` samples for ``seed``: a host int64 tensor (evaluated on the host; no GPU needed)"""
    idx = torch.empty(proj, dtype=torch.int64)
    _check(lib().fewbit_hip_sampled_rows(seed & 0xffffffffffffffff, rows, proj, idx.data_ptr()))
    return idx


def xoshiro128pp(state, n: int):
    """n outputs of xoshiro128++ from `state` (4 words) on the host, as the kernels evaluate it; returns (outputs, new state)"""
    st = (ctypes.c_uint32 * 4)(*state)
    o = (ctypes.c_uint32 * max(n, 1))()
    lib().fewbit_hip_xoshiro128pp(st, o, n)
    return tuple(o)[:n], tuple(st)


def philox4x32(counter, key):
    """Philox4x32-10 on the host, as the kernels evaluate it (no GPU needed)"""
    c = (ctypes.c_uint32 * 4)(*counter)
    k = (ctypes.c_uint32 * 2)(*key)
    o = (ctypes.c_uint32 * 4)()
    lib().fewbit_hip_philox4x32(c, k, o)
    return tuple(o)
