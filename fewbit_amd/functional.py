"""``fewbit.functional``: activation functions whose backward pass uses a few-bit quantized derivative.

Mirror of the reference's functional layer (fewbit/functional/activations.py): the same 22 names, the same
keyword interface (``bits`` xor ``borders``+``values``, default 3 bits), the same error types, and the same
dispatch rule -- a GPU tensor goes to ``torch.ops.fewbit.<name>`` (the gfx950 kernels, in place on the input like
the reference op), a host tensor to a plain-PyTorch autograd Function with unpacked codes.  What differs is
listed in DESIGN.md section 6 ("defects that are not reproduced"): the host path computes the right forward, the 1-bit family works
on host tensors, modules may pass ``bits=`` to 1-bit functions, and a missing native library is an error for
GPU tensors instead of a silent Python fallback.
"""
import inspect
from inspect import Parameter, Signature
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .store import store

# Stepwise (piecewise-linear, exact 1-bit derivative) and continuous activation functions.
STEPWISE = ('hardshrink', 'hardsigmoid', 'hardtanh', 'leaky_relu', 'relu', 'relu6', 'softshrink', 'stepwise',
            'threshold')
CONTINOUS = ('celu', 'elu', 'gelu', 'hardswish', 'logsigmoid', 'mish', 'selu', 'sigmoid', 'silu', 'softplus',
             'softsign', 'tanh', 'tanhshrink')

__all__ = STEPWISE + CONTINOUS + ('store', )

BITS_DEFAULT = 3

# extra positional parameters of each function, as (name, default); Parameter.empty = required
_EXTRA: Dict[str, Tuple[Tuple[str, object], ...]] = {
    'celu': (('alpha', 1.0), ),
    'elu': (('alpha', 1.0), ),
    'softplus': (('beta', 1.0), ('threshold', 20.0)),
    'hardshrink': (('lambd', 0.5), ),
    'hardtanh': (('min_val', -1.0), ('max_val', 1.0)),
    'leaky_relu': (('negative_slope', 0.01), ),
    'softshrink': (('lambd', 0.5), ),
    'threshold': (('threshold', Parameter.empty), ('value', Parameter.empty)),
}


def _torch_impl(name: str) -> Callable:
    if name in ('sigmoid', 'tanh'):
        return getattr(torch, name)
    return getattr(F, name)


def _native_op(name: str):
    """``torch.ops.fewbit.<name>`` or a loud failure: GPU tensors are never computed on the host."""
    from . import native_loaded, native_error
    if not native_loaded():
        raise RuntimeError(f'fewbit: native library is not loaded ({native_error()}); GPU tensors need '
                           'fewbit_amd/libfewbit.so -- build it with `make -C fewbit_amd/csrc`')
    return getattr(torch.ops.fewbit, name)


_OVERLOADS: Dict[str, Callable] = {}


def _native_overload(name: str) -> Callable:
    """``torch.ops.fewbit.<name>.default`` (cached; skips the overload resolution of the packet on every call)."""
    op = _OVERLOADS.get(name)
    if op is None:
        op = _OVERLOADS[name] = _native_op(name).default
    return op


# enum order of include/fewbit_hip.h, for the out-of-place operators
_CONTINUOUS_ID = {n: i for i, n in enumerate(CONTINOUS[:0] + ('celu', 'elu', 'gelu', 'hardswish', 'logsigmoid', 'mish', 'selu',
                                                              'sigmoid', 'silu', 'softplus', 'softsign', 'tanh',
                                                              'tanhshrink', 'identity'))}
_STEPWISE_ID = {n: i for i, n in enumerate(('hardshrink', 'hardsigmoid', 'hardtanh', 'leaky_relu', 'relu', 'relu6',
                                            'softshrink', 'threshold'))}


def _gpu_continuous(name: str, input, inner, levels, extra):
    """In place like the reference op -- except on a view (say, the reshaped output of nn.Linear), where an in-place
    op would make autograd rebase the view (CopySlices: a zero-fill plus four full-size copies in backward); there the
    same kernel writes a fresh tensor and the view's base is simply released."""
    if input._is_view() or not input.is_contiguous():
        p = tuple(extra) + (0.0, ) * (2 - len(extra))
        _native_op(name)  # loud failure if the library is missing
        # (a strided view -- chunk(2, -1) of a GEGLU, a transpose, channels_last -- is gathered first: the kernels
        # stream flat memory)
        return torch.ops.fewbit.continuous_out(input.contiguous(), inner, levels, _CONTINUOUS_ID[name], *p)
    return _native_op(name)(input, inner, levels, *extra)


class _HostQuantized(torch.autograd.Function):
    """Host-tensor path: codes kept one per byte (reference: FallbackFunc, fewbit/functional/activations.py:89-129)."""

    @staticmethod
    def forward(ctx, impl, input, borders, levels, *args):
        if borders.numel() + 1 != levels.numel():
            raise ValueError('Size of `borders` should be lesser than size of `levels` by one.')
        if levels.numel() > 256:                   # codes are kept in one byte each; the GPU operators have the same limit
            raise ValueError(f'Maximal number of levels is 256, got {levels.numel()}.')
        key = input.detach().float().contiguous()
        fold = getattr(impl, 'fold', None)
        if fold is not None:                       # even-parity fold of a custom table: search |x - shift_x|
            key = (key - fold).abs()
        state = torch.searchsorted(borders.float().contiguous(), key).to(torch.uint8)
        ctx.save_for_backward(state, levels)
        ctx.nargs = 4 + len(args)
        return impl(input, *args)

    @staticmethod
    def backward(ctx, grad_output):
        state, levels = ctx.saved_tensors
        return (None, levels[state.long()] * grad_output) + (None, ) * (ctx.nargs - 2)


class _FoldedIdentity:
    """Forward of a custom table (identity) that asks the host path to search ``|x - fold|``."""

    def __init__(self, fold: float):
        self.fold = fold

    def __call__(self, t):
        return t.clone()


def _bind_extra(name: str, args: tuple, kwargs: dict) -> tuple:
    spec = _EXTRA.get(name, ())
    if len(args) > len(spec):
        raise TypeError(f'{name}() takes at most {1 + len(spec)} positional arguments but {1 + len(args)} were given')
    values = list(args)
    for pname, default in spec[len(args):]:
        if pname in kwargs:
            values.append(kwargs.pop(pname))
        elif default is Parameter.empty:
            raise TypeError(f"{name}() missing required argument: '{pname}'")
        else:
            values.append(default)
    for pname, _ in spec[:len(args)]:
        if pname in kwargs:
            raise TypeError(f"{name}() got multiple values for argument '{pname}'")
    if kwargs:
        raise TypeError(f"{name}() got an unexpected keyword argument '{next(iter(kwargs))}'")
    return tuple(values)


def _signature(name: str, continuous: bool) -> Signature:
    params = [Parameter('input', Parameter.POSITIONAL_OR_KEYWORD, annotation=torch.Tensor)]
    params += [Parameter(p, Parameter.POSITIONAL_OR_KEYWORD, default=d, annotation=float) for p, d in _EXTRA.get(name, ())]
    params.append(Parameter('bits', Parameter.KEYWORD_ONLY, default=None, annotation=Optional[int]))
    if continuous:
        params.append(Parameter('borders', Parameter.KEYWORD_ONLY, default=None, annotation=Optional[torch.Tensor]))
        params.append(Parameter('values', Parameter.KEYWORD_ONLY, default=None, annotation=Optional[torch.Tensor]))
    return Signature(params, return_annotation=torch.Tensor)


def _make_continuous(name: str) -> Callable:
    impl = _torch_impl(name)

    def fn(input, *args, bits=None, borders=None, values=None, **kwargs):
        extra = _bind_extra(name, args, kwargs)
        use_builtin = bits is not None
        use_custom = borders is not None and values is not None
        if use_builtin and use_custom:
            raise ValueError('Either `bits` or `borders` and `values` should be scpecifed not both.')
        if use_builtin or not use_custom:
            inner, levels = store.get_inner(name, bits or BITS_DEFAULT, input.device, input.dtype)
        else:
            inner = borders[1:-1].to(input)
            levels = values.to(input)
        if input.device.type == 'cuda':
            return _gpu_continuous(name, input, inner, levels, extra)
        return _HostQuantized.apply(impl, input, inner, levels, *extra)

    fn.__name__ = fn.__qualname__ = name
    fn.__doc__ = (f'Few-bit ``{name}``: forward as :func:`torch.nn.functional.{name}`; backward multiplies the incoming '
                  'gradient by a piecewise-constant approximation of the derivative whose bucket index was\n'
                  'stored with ``bits`` bits per element (default 3).\n\n'
                  'In-place rule (GPU): a tensor that owns its memory is overwritten and returned, like the reference op '
                  '(``Tensor(a!)``); a VIEW or a\nnon-contiguous tensor is left intact and a fresh tensor is returned '
                  '(an in-place write through a view would make autograd rebase it).  Host tensors are never modified.\n\n'
                  'Either ``bits`` (built-in table) or ``borders`` and ``values`` (custom table; ``borders`` with '
                  'both outer sentinels) may be given.')
    fn.__signature__ = _signature(name, True)
    return fn


def _make_stepwise1(name: str) -> Callable:
    impl = _torch_impl(name)

    def fn(input, *args, bits=None, **kwargs):
        # `bits` is accepted and ignored: these functions have an exact 1-bit state (generated modules pass it)
        extra = _bind_extra(name, args, kwargs)
        if input.device.type == 'cuda':
            if input._is_view() or not input.is_contiguous():                 # see _gpu_continuous
                p = tuple(extra) + (0.0, ) * (2 - len(extra))
                _native_op(name)
                return torch.ops.fewbit.stepwise1_out(input.contiguous(), _STEPWISE_ID[name], *p)
            return _native_op(name)(input, *extra)
        return impl(input, *extra)

    fn.__name__ = fn.__qualname__ = name
    fn.__doc__ = (f'Few-bit ``{name}``: same values and gradients as :func:`torch.nn.functional.{name}`, but only one '
                  'bit per element is saved for backward.  GPU tensors that own their memory are overwritten in place; views and '
                  'non-contiguous tensors are left intact (a fresh tensor is returned).')
    fn.__signature__ = _signature(name, False)
    return fn


def stepwise(input: torch.Tensor, borders: torch.Tensor, levels: torch.Tensor, parity: Optional[bool] = None,
             shift: Optional[Tuple[float, float]] = None) -> torch.Tensor:
    """Identity forward with a custom stepwise derivative: ``borders`` are the inner borders (one fewer than
    ``levels``).

    ``parity``/``shift`` are declared by the reference and implemented nowhere in it (fewbit/fewbit.cc:37,
    fewbit/functional/activations.py:137-139 raises NotImplementedError); here, with ``shift = (sx, sy)`` and the
    table given on the half line ``t = |x - sx| >= 0`` (what ``approximate(parity=True, domain=(0, x_max))`` makes):

    * ``parity=True``  -- the step function is even about ``sx``: ``level = levels[#{b < |x - sx|}]``; the fold
      happens inside the kernel, so k bits address 2^k half-line levels;
    * ``parity=False`` -- odd about ``(sx, sy)``: ``level = l'`` right of ``sx`` and ``2*sy - l'`` left of it, i.e.
      the plain table mirrored to twice its size (one more bit: the sign).
    """
    if parity is None:
        if shift is not None:
            raise ValueError('`shift` needs a `parity`.')
    else:
        sx, sy = (0.0, 0.0) if shift is None else (float(shift[0]), float(shift[1]))
        if input.device.type == 'cuda':
            if input._is_view() or not input.is_contiguous():
                return _native_op('stepwise_folded_out')(input.contiguous(), borders.to(input), levels.to(input), bool(parity), sx, sy)
            return _native_op('stepwise_folded')(input, borders.to(input), levels.to(input), bool(parity), sx, sy)
        if parity:
            return _HostQuantized.apply(_FoldedIdentity(sx), input, borders.to(input), levels.to(input))
        if levels.numel() > 128:                   # as torch.ops.fewbit.stepwise_folded (torch_ops.cpp)
            raise ValueError(f'An odd-parity table mirrors to twice its size: at most 128 levels, got {levels.numel()}.')
        bf, lf = borders.float(), levels.float()
        full_b = torch.cat([sx - bf.flip(0), bf.new_full((1, ), sx), bf + sx]).to(input)
        full_l = torch.cat([2.0 * sy - lf.flip(0), lf]).to(input)
        return _HostQuantized.apply(lambda t: t.clone(), input, full_b, full_l)
    if input.device.type == 'cuda':
        if input._is_view() or not input.is_contiguous():
            _native_op('stepwise')
            return torch.ops.fewbit.continuous_out(input.contiguous(), borders.to(input), levels.to(input), _CONTINUOUS_ID['identity'], 0.0, 0.0)
        return _native_op('stepwise')(input, borders.to(input), levels.to(input))
    return _HostQuantized.apply(lambda t: t.clone(), input, borders.to(input), levels.to(input))


for _name in CONTINOUS:
    globals()[_name] = _make_continuous(_name)
for _name in STEPWISE:
    if _name != 'stepwise':
        globals()[_name] = _make_stepwise1(_name)
del _name

# Randomized linear layers and gradient-capture helpers live in the same namespace in the reference
# (fewbit/functional/__init__.py:14-18).
from .linear import linear_crs, linear_grp, linear_randomized  # noqa: E402,F401
from .variance import GradientStorage, catch_gradients  # noqa: E402,F401
