"""Gradient-capture utilities and the variance estimator for randomized linear layers.

Counterpart of the reference's ``fewbit/functional/variance.py`` (``GradientStorage``, ``catch_gradients``) and
``fewbit/modules/variance.py`` (``VarianceEstimator``): same names and call shapes, own implementation.  For a linear
layer with input rows ``X`` (B x n) and output-gradient rows ``G`` (B x m) the exact weight gradient is ``G^T X``; the
module reports, per backward pass,

* ``corr     = (||X^T G||_F / (||X||_F ||G||_F))^2``
* ``var_sgd  = B/(B-1) * sum_b ||x_b||^2 ||g_b||^2 - ||X^T G||_F^2 / (B-1)``     (variance of the SGD estimate)
* ``var_rmm  = (||X||_F^2 ||G||_F^2 - ||X^T G||_F^2) / B_proj``                   (added by the random projection)

(definitions of fewbit/modules/variance.py:17-46; arXiv:2201.13195, section 3).
"""
from typing import Callable, Optional

import torch

from .linear import projection_dim

__all__ = ('GradientStorage', 'catch_gradients', 'VarianceEstimator', 'estimate_correlation', 'estimate_variance_sgd',
           'estimate_variance_rmm')


class GradientStorage:
    """Keeps a copy of a module's last input and of the gradient that reached its output."""

    def __init__(self):
        self.input = None
        self.grad_output = None

    def forward(self, input: torch.Tensor) -> None:
        self.input = input.detach().clone()

    def backward(self, grad_output: torch.Tensor) -> None:
        self.grad_output = grad_output.detach().clone()
        self.postprocess()

    def postprocess(self) -> None:
        """Hook for subclasses: called once both tensors of a step are in."""


class _Catch(torch.autograd.Function):

    @staticmethod
    def forward(ctx, input: torch.Tensor, storage: GradientStorage) -> torch.Tensor:
        ctx.storage = storage
        return input.view_as(input)

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor):
        ctx.storage.backward(grad_output)
        return grad_output, None


def catch_gradients(input: torch.Tensor, storage: GradientStorage) -> torch.Tensor:
    """Identity whose backward hands the passing gradient to ``storage.backward``."""
    return _Catch.apply(input, storage)


def estimate_correlation(input: torch.Tensor, output: torch.Tensor) -> torch.Tensor:
    cross = torch.linalg.norm(input.T @ output)
    return (cross / (torch.linalg.norm(input) * torch.linalg.norm(output)))**2


def estimate_variance_sgd(input: torch.Tensor, output: torch.Tensor, bs: Optional[int] = None) -> torch.Tensor:
    bs = bs or input.shape[0]
    rows = (input * input).sum(dim=1) @ (output * output).sum(dim=1)
    cross = torch.linalg.norm(input.T @ output)**2
    return rows * (bs / (bs - 1)) - cross / (bs - 1)


def estimate_variance_rmm(input: torch.Tensor, output: torch.Tensor, bs_proj: Optional[int] = None) -> torch.Tensor:
    bs_proj = bs_proj or input.shape[0]
    cross = torch.linalg.norm(input.T @ output)**2
    return (torch.linalg.norm(input)**2 * torch.linalg.norm(output)**2 - cross) / bs_proj


class _VarianceState(GradientStorage):

    def __init__(self, callback: Optional[Callable] = None):
        super().__init__()
        self.callback = callback
        self.step = 0
        self.variance = None
        self.bs = None
        self.bs_proj = None

    def postprocess(self) -> None:
        if self.input is None or self.grad_output is None:
            return
        x = self.input.reshape(-1, self.input.shape[-1]).float()
        g = self.grad_output.reshape(-1, self.grad_output.shape[-1]).float()
        corr = estimate_correlation(x, g)
        var_sgd = estimate_variance_sgd(x, g, self.bs)
        var_rmm = estimate_variance_rmm(x, g, self.bs_proj)
        if callable(self.callback):
            self.callback(corr, var_sgd, var_rmm, self.step)
        self.step += 1
        self.variance = (corr, var_sgd, var_rmm)


class VarianceEstimator(torch.nn.Module):
    """Wraps a randomized linear layer (anything with ``proj_dim*`` attributes) and evaluates the three quantities
    above on every backward pass; ``callback(corr, var_sgd, var_rmm, step)`` receives them, ``.variance`` keeps the
    last triple.  (The reference stores ``var_sgd`` twice in ``.variance``, fewbit/modules/variance.py:77; here the
    third entry is ``var_rmm``.)"""

    def __init__(self, model: torch.nn.Module, callback: Optional[Callable] = None):
        super().__init__()
        self.model = model
        self.state = _VarianceState(callback)

    @property
    def variance(self):
        return self.state.variance

    def forward(self, input: torch.Tensor, *args, **kwargs):
        rows = input.numel() // input.shape[-1]
        self.state.bs = rows
        self.state.bs_proj = projection_dim(rows, getattr(self.model, 'proj_dim_ratio', None),
                                            getattr(self.model, 'proj_dim', None),
                                            getattr(self.model, 'proj_dim_max', None),
                                            getattr(self.model, 'proj_dim_min', None))
        self.state.forward(input)
        output = self.model(input, *args, **kwargs)
        if isinstance(output, tuple):
            return (catch_gradients(output[0], self.state), ) + tuple(output[1:])
        return catch_gradients(output, self.state)
