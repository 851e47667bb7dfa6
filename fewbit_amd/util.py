"""Helpers around the hot path: module-tree rewriting and saved-tensor accounting
(reference: fewbit/util.py -- ``map_module``, ``memory_usage_hooks``, ``estimate_memory_usage``, ``teniter``,
``traverse``, ``convert_linear``)."""
import re
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Any, Callable, Iterator, Optional

import torch

__all__ = ['map_module', 'memory_usage_hooks', 'HookedMemoryUsage', 'estimate_memory_usage', 'teniter', 'traverse',
           'convert_linear']


def traverse(variable: torch.Tensor, callback: Callable[[Any, torch.Tensor, bool], Any]) -> None:
    """Walk the backward graph below ``variable`` and call ``callback(node, tensor, saved)`` for every tensor a node
    holds: what it saved for backward (``saved=True``: built-in nodes expose ``_saved_*`` attributes, custom Functions
    ``saved_tensors``) and the leaf it accumulates into (``saved=False``).  A tensor can show up several times."""
    seen, stack = set(), [variable.grad_fn]
    while stack:
        node = stack.pop()
        if node is None or node in seen:
            continue
        seen.add(node)
        held = list(getattr(node, 'saved_tensors', ()))
        for attr in dir(node):
            if attr.startswith('_saved_'):
                try:
                    val = getattr(node, attr)
                except RuntimeError:            # already freed by a backward pass
                    continue
                held += [val] if torch.is_tensor(val) else [t for t in val if torch.is_tensor(t)] if isinstance(val, tuple) else []
        for ten in held:
            callback(node, ten, True)
        if hasattr(node, 'variable'):
            callback(node, node.variable.data, False)
        stack.extend(child for child, _ in reversed(getattr(node, 'next_functions', ())))


def teniter(variable: torch.Tensor, include_ordinary: bool = True, include_saved: bool = False) -> Iterator[torch.Tensor]:
    """Tensors reachable from ``variable`` through ``grad_fn``: leaves (``include_ordinary``) and/or tensors saved for
    backward (``include_saved``).  Like the reference, identity is the Python object, so a buffer saved by two nodes
    counts twice -- the same accounting ``memory_usage_hooks`` does."""
    found = {}

    def note(_node, ten, saved):
        _, ordinary, was_saved = found.get(id(ten), (ten, False, False))
        found[id(ten)] = (ten, ordinary or not saved, was_saved or saved)

    traverse(variable, note)
    for ten, ordinary, saved in found.values():
        if (include_ordinary and ordinary) or (include_saved and saved):
            yield ten


def estimate_memory_usage(variable: torch.Tensor, saved_only: bool = False) -> int:
    """Bytes held by the tensors reachable from ``variable``: the leaves, or with ``saved_only`` what autograd saved."""
    picked = teniter(variable, False, True) if saved_only else teniter(variable, True, False)
    return sum(t.numel() * t.element_size() for t in picked)


def convert_linear(module: torch.nn.Module, ctor, **kwargs) -> torch.nn.Module:
    """``nn.Linear`` -> ``ctor(in_features, out_features, bias, device, dtype, **kwargs)`` sharing the parameters
    (e.g. ``ctor=fewbit.RandomizedLinear``); anything else is returned as is.  For use with :func:`map_module`."""
    if not isinstance(module, torch.nn.Linear):
        return module
    layer = ctor(in_features=module.in_features, out_features=module.out_features, bias=module.bias is not None,
                 device=module.weight.device, dtype=module.weight.dtype, **kwargs)
    layer.weight = torch.nn.Parameter(module.weight)
    if layer.bias is not None:
        layer.bias = torch.nn.Parameter(module.bias)
    return layer


@dataclass
class HookedMemoryUsage:
    forward: Optional[int] = None   # bytes packed by forward passes (saved for backward)
    backward: Optional[int] = None  # bytes unpacked by backward passes

    @property
    def value(self) -> Optional[int]:
        return self.backward or self.forward


@contextmanager
def memory_usage_hooks() -> Iterator[HookedMemoryUsage]:
    """Count the bytes autograd saves (and later reads back) inside the ``with`` block."""
    usage = HookedMemoryUsage()

    def pack(t: torch.Tensor):
        usage.forward = (usage.forward or 0) + t.numel() * t.element_size()
        return t

    def unpack(t: torch.Tensor):
        usage.backward = (usage.backward or 0) + t.numel() * t.element_size()
        return t

    with torch.autograd.graph.saved_tensors_hooks(pack, unpack):
        yield usage


def map_module(root: torch.nn.Module, func: Callable[[torch.nn.Module, str], torch.nn.Module],
               patt: Optional[str] = None) -> torch.nn.Module:
    """Apply ``func(module, path)`` bottom-up to every module whose '/'-separated path matches ``patt`` (regex,
    default: all) and splice the returned modules into the tree.  Returns the (possibly replaced) root."""
    pattern = re.compile(patt or r'.*')

    def visit(node: torch.nn.Module, path: str) -> torch.nn.Module:
        for name, child in list(node.named_children()):
            new = visit(child, f'{path}/{name}')
            if new is not child:
                setattr(node, name, new)
        if pattern.match(path or '/'):
            out = func(node, path or '/')
            if not isinstance(out, torch.nn.Module):
                raise ValueError('Mapped result should be toch.nn.Module type.')
            return out
        return node

    return visit(root, '')
