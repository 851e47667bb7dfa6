"""Splitting one flattened activation across GPUs: no data-path collective, every shard is self-contained.

The path is element-wise and every group of 8 elements maps to exactly k state bytes, so a tensor can be cut at
any multiple of 8 elements.  Cuts are placed at multiples of ``ALIGN`` = 512 elements (one wave tile of 64 lanes
x 8 elements; also keeps every shard's state dword-aligned for any k).  Shard r owns elements [begin, end) and
state bytes [k*begin/8, k*ceil(end/8)): concatenating the shards' outputs reproduces the single-GPU result
bit for bit.
"""
from typing import Tuple

ALIGN = 512

__all__ = ['ALIGN', 'shard_range', 'state_range']


def shard_range(n: int, world: int, rank: int, align: int = ALIGN) -> Tuple[int, int]:
    """Element range [begin, end) of ``rank`` when ``n`` elements are split over ``world`` ranks."""
    if not 0 <= rank < world:
        raise ValueError(f'rank {rank} outside [0, {world})')
    blocks = -(-n // align)                      # number of aligned blocks, last one may be ragged
    base, extra = divmod(blocks, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    begin = min(first * align, n)
    end = min((first + count) * align, n)
    return begin, end


def state_range(begin: int, end: int, bits: int) -> Tuple[int, int]:
    """Byte range of the packed state belonging to elements [begin, end); ``begin`` must be a multiple of 8."""
    if begin % 8:
        raise ValueError('shards must start on a multiple of 8 elements')
    return bits * begin // 8, bits * (-(-end // 8))
