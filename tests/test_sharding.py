"""Multi-GPU path on CPU: two gloo ranks each process their own shard (oracle standing in for the device) and the
concatenation must equal the unsharded result bit for bit -- no collective touches the data path; the gather here
is test plumbing only."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fewbit_amd.sharding import ALIGN, shard_range, state_range


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 511, 512, 513, 4096, 100003, 16777216, 268435456):
        for world in (1, 2, 3, 4, 8):
            pos = 0
            for r in range(world):
                b, e = shard_range(n, world, r)
                assert b == pos and b <= e and (b % ALIGN == 0 or b == n)
                pos = e
            assert pos == n
    assert shard_range(4 * 16384 * 4096, 8, 3) == (3 * 33554432, 4 * 33554432)     # BASELINE config 4
    assert state_range(1024, 2048, 3) == (384, 768)
    with pytest.raises(ValueError):
        state_range(4, 16, 3)


def _worker(rank, world, port, n, bits, dtype):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import oracle
    from fewbit_amd.store import store
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        x = (torch.randn(n, generator=g) * 1.5).to(dtype)              # every rank builds the same full tensor ...
        gy = torch.randn(n, generator=g).to(dtype)
        borders, levels = store.get('gelu', bits, 'cpu', dtype)
        b, e = shard_range(n, world, rank)                              # ... and touches only its own shard
        y, state, k = oracle.quantize('gelu', x[b:e], borders[1:-1])
        gx = oracle.quantize_backward(gy[b:e], state, levels)
        sb, se = state_range(b, e, k)
        assert state.numel() == se - sb
        parts = [None] * world
        dist.all_gather_object(parts, (b, e, y, state, gx))
        if rank == 0:
            y_full, state_full, _ = oracle.quantize('gelu', x, borders[1:-1])
            gx_full = oracle.quantize_backward(gy, state_full, levels)
            view = torch.int16 if dtype != torch.float32 else torch.int32
            assert torch.equal(torch.cat([p[2] for p in parts]).view(view), y_full.view(view))
            assert torch.equal(torch.cat([p[3] for p in parts]), state_full)
            assert torch.equal(torch.cat([p[4] for p in parts]).view(view), gx_full.view(view))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n,bits,dtype', [(512 * 7 + 5, 3, torch.bfloat16), (4096, 2, torch.float16), (1000, 4, torch.float32)])
def test_two_ranks_reproduce_single_rank(n, bits, dtype):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, n, bits, dtype), nprocs=2, join=True)
