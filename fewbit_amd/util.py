"""Helpers around the hot path: module-tree rewriting and saved-tensor accounting
(reference: ``map_module`` and ``memory_usage_hooks``, fewbit/util.py:126-187)."""
import re
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Callable, Iterator, Optional

import torch

__all__ = ['map_module', 'memory_usage_hooks', 'HookedMemoryUsage']


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
