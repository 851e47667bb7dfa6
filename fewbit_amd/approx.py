"""Offline construction of the quantization tables: the best (L2) piecewise-constant approximation of an
activation's derivative with a given number of levels.

Same interface and the same alternating procedure as the reference's table generator (fewbit/approx.py:64-154, the
producer of fewbit/data/builtin.npz through `fewbit quantize`, fewbit/cli.py:60-124), so that seeded runs reproduce
the built-in tables; see tests/test_approx.py.  For borders b_0 < ... < b_m and levels l_1..l_m on (b_{i-1}, b_i]:

    optimal level for fixed borders : l_i = (F(b_i) - F(b_{i-1})) / (b_i - b_{i-1}),   F' = f   (mean of f)
    stationarity in an inner border : f(b_i) = (l_i + l_{i+1}) / 2
    update (fixed step)             : b_i <- b_i - 2 (l_{i+1} - l_i) (f(b_i) - (l_i + l_{i+1}) / 2)
"""
from io import StringIO
from typing import Any, Callable, Dict, Tuple, Union

import numpy as np
from numpy.typing import ArrayLike, NDArray

__all__ = ('StepWiseFunction', 'approximate', 'estimate_error', 'stepwise')

RandomState = Union[None, int, ArrayLike, np.random.RandomState]
VectorizedFn = Callable[[ArrayLike], NDArray]


class StepWiseFunction:
    """Piecewise-constant function: ``levels[i]`` on ``(borders[i], borders[i+1]]`` (a point on a border takes the
    lower level -- the same rule the kernels use to bucket an input)."""

    def __init__(self, borders: NDArray, levels: NDArray):
        borders, levels = np.asarray(borders), np.asarray(levels)
        if borders.ndim != 1 or levels.ndim != 1 or borders.size != levels.size + 1:
            raise ValueError('expected 1-d `borders` one longer than 1-d `levels`')
        self.borders = borders
        self.levels = levels
        self.card = levels.size
        self.steps = np.concatenate([levels[:1], np.diff(levels)])

    def __call__(self, xs: ArrayLike) -> NDArray:
        xs = np.asarray(xs)
        return self.levels[np.searchsorted(self.borders[1:-1], xs, side='left')]

    def __repr__(self) -> str:
        more = ', ...' if self.borders.size > 2 else ''
        bs = ', '.join(f'{x:e}' for x in self.borders[:2]) + more
        ls = ', '.join(f'{x:e}' for x in self.levels[:2]) + more
        return f'<StepWiseFunction nosteps={self.card} borders=[{bs}] levels=[{ls}]>'

    def __str__(self) -> str:
        buf = StringIO()
        for i, level in enumerate(self.levels):
            print(f'[{i}] [{self.borders[i]:+8.3f}, {self.borders[i + 1]:+8.3f}) => {level:e}', file=buf)
        return buf.getvalue()


def stepwise(xs: ArrayLike, ys: ArrayLike) -> StepWiseFunction:
    return StepWiseFunction(np.asarray(xs), np.asarray(ys))


def is_sorted(xs: ArrayLike, scale: float = 0.0) -> bool:
    return bool(np.all(np.diff(xs) > scale))


def _mean_levels(fn_prim: VectorizedFn, borders: NDArray) -> NDArray:
    return np.diff(fn_prim(borders)) / np.diff(borders)


def approximate(fn: VectorizedFn, fn_prim: VectorizedFn, cardinality: int, domain: Tuple[float, float] = (-100.0, 100.0),
                parity: bool = False, max_iters: int = 10000, beps: float = 1e-4, leps: float = 1e-4,
                random_state: RandomState = None) -> Tuple[StepWiseFunction, Dict[str, Any]]:
    """Stepwise approximation of ``fn`` (whose primitive is ``fn_prim``) with ``cardinality`` levels on ``domain``.

    ``parity=True`` fits the non-negative half only (``domain[0]`` must be 0).  Returns the function and a dict
    ``{'status': 'converged' | 'not-converged' | 'failed', 'noiters', 'bs_diff', 'ls_diff'}``.
    """
    lo, hi = domain
    if parity and lo != 0.0:
        raise ValueError('parity fits need a domain starting at 0')
    rng = np.random.RandomState(random_state)

    borders = np.empty(cardinality + 1)
    borders[0], borders[-1] = lo, hi
    for _ in range(16):                                   # initial lattice: N(0, 1.5) draws, at least 1e-3 apart
        draw = rng.normal(0.0, 1.5, cardinality - 1)
        borders[1:-1] = np.abs(draw) if parity else draw
        borders.sort()
        if is_sorted(borders, 1e-3):
            break
    else:
        raise RuntimeError('Failed to generate initial lattice!')

    levels = _mean_levels(fn_prim, borders)
    status, it, bs_diff, ls_diff = 'not-converged', 0, np.inf, np.inf
    for it in range(max_iters):
        inner = borders[1:-1]
        step = -2.0 * np.diff(levels) * (fn(inner) - 0.5 * (levels[:-1] + levels[1:]))
        borders[1:-1] = inner + step
        step_norm = np.linalg.norm(step)
        bs_diff = step_norm / np.linalg.norm(borders)
        if step_norm < beps:
            status = 'converged'
            break
        new_levels = _mean_levels(fn_prim, borders)
        ls_diff = np.linalg.norm(new_levels - levels) / np.linalg.norm(levels)
        levels = new_levels
        if ls_diff < leps:
            status = 'converged'
            break
        if not is_sorted(borders):                        # crossing borders never recover
            status = 'failed'
            break
    return stepwise(borders, levels), {'status': status, 'noiters': it, 'bs_diff': bs_diff, 'ls_diff': ls_diff}


def estimate_error(fn: VectorizedFn, fn_approx: StepWiseFunction, dx: float):
    """Integrated squared error of the approximation, total and per piece (Simpson's rule on a grid of step ~dx)."""
    from scipy.integrate import simpson
    errs = np.empty(fn_approx.card)
    for i, level in enumerate(fn_approx.levels):
        a, b = fn_approx.borders[i:i + 2]
        xs = np.linspace(a, b, min(1024**2, int((b - a) / dx)))
        errs[i] = simpson((fn(xs) - level)**2, x=xs)
    return errs.sum(), errs
