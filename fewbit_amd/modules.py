"""``fewbit.<Module>``: drop-in ``torch.nn`` activation modules with a few-bit backward.

Mirror of fewbit/modules/activations.py: 21 classes generated from the matching ``torch.nn`` classes (constructor
signature minus ``inplace``/``approximate`` plus keyword-only ``bits``; ``repr`` like ``GELU(bits=3)``) and
``Stepwise`` for a custom table.
"""
import inspect
from inspect import Parameter, Signature
from typing import Optional, Tuple

import torch

from . import functional

STEPWISE = ('Hardshrink', 'Hardsigmoid', 'Hardtanh', 'LeakyReLU', 'ReLU', 'ReLU6', 'Softshrink', 'Stepwise',
            'Threshold')
CONTINOUS = ('CELU', 'ELU', 'GELU', 'Hardswish', 'LogSigmoid', 'Mish', 'SELU', 'Sigmoid', 'SiLU', 'Softplus',
             'Softsign', 'Tanh', 'Tanhshrink')

__all__ = STEPWISE + CONTINOUS

_PARAM_BITS = Parameter('bits', Parameter.KEYWORD_ONLY, default=None, annotation=Optional[int])


class Stepwise(torch.nn.Module):
    """Custom stepwise approximation: identity forward, ``levels[bucket(x)] * grad`` backward.

    :param borders: inner borders of the intervals (or with both outer sentinels, which are then dropped).
    :param levels: value of the derivative on every interval.
    :param parity: ``True``: the table is given on ``t = |x - shift_x| >= 0`` and the step function is even about
                   ``shift_x``; ``False``: odd about ``(shift_x, shift_y)``; ``None``: plain table.
                   See :func:`fewbit.functional.stepwise`.
    :param shift: ``(shift_x, shift_y)``, origin of the symmetry (default ``(0, 0)``).
    """

    def __init__(self, borders: torch.Tensor, levels: torch.Tensor, parity: Optional[bool] = None,
                 shift: Optional[Tuple[float, float]] = None):
        if borders.ndim != 1 or levels.ndim != 1:
            raise ValueError('Exepected number of dimensions of `borders` and `levels` is one.')
        if borders.numel() > levels.numel():
            borders = borders[1:-1]
        if borders.numel() + 1 != levels.numel():
            raise ValueError('Size of `borders` should be lesser than size of `levels` by one.')
        if levels.numel() > 256:
            raise ValueError('Maximal number of step limited to 256.')
        super().__init__()
        self.register_buffer('borders', borders, True)
        self.register_buffer('levels', levels, True)
        self.parity = parity
        self.shift = shift

    def forward(self, xs: torch.Tensor) -> torch.Tensor:
        return functional.stepwise(xs, self.borders, self.levels, self.parity, self.shift)


class BuiltInStepwiseFunction(torch.nn.Module):
    """Base of the generated modules: binds the constructor arguments once, forwards them on every call."""

    _impl_name: str = ''
    _signature: Signature = Signature()

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        ref = getattr(torch.nn, cls.__name__)
        sig = inspect.signature(ref.__init__)
        params = list(sig.parameters.values())
        # torch >= 2 declares some activations as (self, *args, **kwargs): keep `self` only
        if len(params) == 3 and params[1].kind == Parameter.VAR_POSITIONAL and params[2].kind == Parameter.VAR_KEYWORD:
            params = params[:1]
        params = [p for p in params if p.name not in ('approximate', 'inplace')] + [_PARAM_BITS]
        cls._signature = sig.replace(parameters=params)
        cls._impl_name = 'leaky_relu' if cls.__name__ == 'LeakyReLU' else cls.__name__.lower()
        cls._impl = staticmethod(getattr(functional, cls._impl_name))
        accepted = inspect.signature(cls._impl).parameters
        cls._forwarded = tuple(p.name for p in params[1:] if p.name in accepted)

        def __init__(self, *args, **kwargs):
            BuiltInStepwiseFunction.__init__(self, *args, **kwargs)

        __init__.__signature__ = cls._signature
        cls.__init__ = __init__
        cls.__doc__ = (f'Few-bit :class:`torch.nn.{ref.__name__}`: same forward, backward from a ``bits``-bit code per '
                       f'element (default 3; exact 1-bit state for piecewise-linear functions).\n\n'
                       f'    See Also:\n        :class:`torch.nn.{ref.__name__}` -- Original PyTorch implementation.\n')

    def __init__(self, *args, **kwargs):
        super().__init__()
        bound = self._signature.bind(self, *args, **kwargs)
        bound.apply_defaults()
        bound.arguments.pop('self')
        self.reprs = []
        for name, value in bound.arguments.items():
            setattr(self, name, value)
            self.reprs.append(f'{name}={value}')
        # keyword arguments handed to the functional on every forward (only those it knows)
        self.kwargs = {name: bound.arguments[name] for name in self._forwarded}
        # GPU fast path (the per-call signature binding of the functional layer costs as much as a small
        # kernel): the positional extras are bound ONCE here, the table casts are kept per (device, dtype), and forward()
        # calls the operator overload directly
        self._extra = tuple(bound.arguments[p] for p, _ in functional._EXTRA.get(self._impl_name, ()))
        self._tables = {}
        self._tables_version = -1

    def __repr__(self) -> str:
        return f'{type(self).__name__}({", ".join(self.reprs)})'

    def forward(self, xs: torch.Tensor) -> torch.Tensor:
        if xs.device.type != 'cuda':
            return self._impl(xs, **self.kwargs)
        name = self._impl_name
        flat = not xs._is_view() and xs.is_contiguous()           # owns its memory: in place, like the reference op
        if name in functional._STEPWISE_ID:
            if flat:
                return functional._native_overload(name)(xs, *self._extra)
            p = self._extra + (0.0, ) * (2 - len(self._extra))
            return functional._native_overload('stepwise1_out')(xs.contiguous(), functional._STEPWISE_ID[name], *p)
        store = functional.store
        if self._tables_version != store.version:                  # a table was replaced through store.add()
            self._tables.clear()
            self._tables_version = store.version
        key = (xs.device, xs.dtype)
        hit = self._tables.get(key)
        if hit is None:
            hit = self._tables[key] = store.get_inner(name, self.bits or functional.BITS_DEFAULT, xs.device, xs.dtype)
        if flat:
            return functional._native_overload(name)(xs, hit[0], hit[1], *self._extra)
        p = self._extra + (0.0, ) * (2 - len(self._extra))
        return functional._native_overload('continuous_out')(xs.contiguous(), hit[0], hit[1], functional._CONTINUOUS_ID[name], *p)


for _name in __all__:
    if _name not in globals():
        globals()[_name] = type(_name, (BuiltInStepwiseFunction, ), {'__module__': __name__})
del _name
