"""bench.py's own N > 1 launcher (`python bench.py --gpus N` with no torchrun around it) on the CPU: the parent starts N
children, they meet at file barriers, the parent prints ONE JSON line with the aggregate; a failing rank makes it exit
non-zero.  The children here are a stand-in worker (FEWBIT_BENCH_WORKER) that speaks bench.py's worker protocol with made-up
timings -- the real worker needs a GPU and is exercised by `gpurun` (profiles/r03_bench_line_2ranks_selflaunch.json)."""
import json
import os
import subprocess
import sys
import textwrap
import threading
import time

import pytest

from helpers import ROOT

STUB = textwrap.dedent('''
    import argparse, json, os, sys, time
    from pathlib import Path
    sys.path.insert(0, {root!r})
    import bench
    ap = argparse.ArgumentParser()
    for a in ('--gpus', '--steps', '--warmup', '--worker-rank'):
        ap.add_argument(a, type=int)
    ap.add_argument('--config'); ap.add_argument('--settle-ms'); ap.add_argument('--sync-dir'); ap.add_argument('--scaling')
    ap.add_argument('--digests', action='store_true')
    args = ap.parse_args()
    mode = os.environ.get('STUB_MODE', 'ok')
    rank, world = args.worker_rank, args.gpus
    if mode == 'die_early' and rank == 1:
        sys.exit(3)
    if mode == 'raise' and rank == 2:
        try:
            raise ValueError('the reason rank 2 died')
        except ValueError:
            import traceback
            (Path(args.sync_dir) / f'error.{{rank}}').write_text(traceback.format_exc())
            raise
    sync = bench.FileSync(args.sync_dir, rank, world, timeout=30.0)
    sync.barrier('start')
    wall = 1e-3 * (1.0 + 0.1 * rank)             # rank r "takes" (1 + 0.1 r) ms for the K steps
    sync.barrier('stop')
    if mode == 'die_late' and rank == world - 1:
        sys.exit(4)
    from fewbit_amd.sharding import shard_range
    n = 4096 * 4096
    begin, end = shard_range(n, world, rank) if args.scaling == 'strong' else (0, n)
    res = dict(rank=rank, device=0 if mode == 'same_device' else rank, n_devices=world, wall_s=wall, event_s=0.9 * wall,
               own_wall_s=wall, own_event_s=0.9 * wall, settle_steps=0, set_bytes=140509184, elements=end - begin, span=[begin, end])
    if rank == 0:
        res.update(fwd_us=12.0, bwd_us=11.0, steady_step_us=23.5,
                   kernels=[dict(kernel='stub_fwd', blocks=1, threads=1, blocks_per_cu=1, chunk=0, u=1, bits=3),
                            dict(kernel='stub_bwd', blocks=1, threads=1, blocks_per_cu=1, chunk=0, u=1, bits=3)])
    tmp = Path(args.sync_dir) / f'result.{{rank}}.json.tmp'
    tmp.write_text(json.dumps(res))
    tmp.rename(Path(args.sync_dir) / f'result.{{rank}}.json')
''')


def _run(tmp_path, mode, gpus=4, steps=20, extra=()):
    stub = tmp_path / 'stub_worker.py'
    stub.write_text(STUB.format(root=str(ROOT)))
    env = dict(os.environ, FEWBIT_BENCH_WORKER=str(stub), STUB_MODE=mode)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    return subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', str(gpus), '--steps', str(steps), '--warmup', '5',
                           '--launch-timeout', '60', *extra], env=env, capture_output=True, text=True, timeout=120)


def test_parent_starts_one_child_per_gpu_and_prints_one_line(tmp_path):
    r = _run(tmp_path, 'ok', gpus=4, steps=20)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 4 and line['steps'] == 20 and line['warmup'] == 5 and line['scaling'] == 'weak'
    # aggregate = the units ALL ranks processed / the slowest rank's time
    slowest = 1e-3 * 1.3
    assert line['ms_per_step'] == pytest.approx(slowest / 20 * 1e3, rel=1e-3)
    assert line['value'] == pytest.approx(4 * 20 * 146800640 / slowest / 2**30, rel=1e-3)
    assert line['per_gpu_us_per_step'] == [pytest.approx(1e3 * (1 + 0.1 * r_) / 20, rel=1e-3) for r_ in range(4)]
    assert 'started 4 child processes itself' in line['config']['launcher']
    assert 'no collectives' in line['config']['parallelism']


@pytest.mark.parametrize('mode', ('die_early', 'die_late'))
def test_a_failed_rank_makes_the_parent_fail(tmp_path, mode):
    t0 = time.time()
    r = _run(tmp_path, mode, gpus=3)
    assert r.returncode != 0
    assert r.stdout.strip() == ''                     # no result line from a broken run
    assert 'failed' in r.stderr
    assert time.time() - t0 < 60                      # the survivors are released by the abort flag, not by their timeout


def test_eight_children_weak(tmp_path):
    r = _run(tmp_path, 'ok', gpus=8, steps=20)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip())
    assert line['n_gpus'] == 8 and line['per_gpu_device'] == list(range(8)) and 'shared_gpu' not in line
    assert line['value'] == pytest.approx(8 * 20 * 146800640 / (1e-3 * 1.7) / 2**30, rel=1e-3)
    assert line['pct_of_hbm_roofline'] == pytest.approx(100 * 8 * 20 * 146800640 / (1e-3 * 1.7) / 1e9 / (8 * 8000.0), rel=1e-3)


@pytest.mark.parametrize('gpus', (2, 4, 8))
def test_strong_scaling_line_is_one_tensor_over_n_ranks(tmp_path, gpus):
    """--scaling strong: the bytes of ONE 4096x4096 tensor / the slowest rank; every rank reports its slice."""
    r = _run(tmp_path, 'ok', gpus=gpus, steps=20, extra=('--scaling', 'strong'))
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip())
    n = 4096 * 4096
    assert line['scaling'] == 'strong' and line['n_gpus'] == gpus
    assert sum(line['per_gpu_elements']) == n and line['config']['elements_total'] == n
    assert line['config']['elements_per_gpu'] == n // gpus == max(line['per_gpu_elements'])
    assert line['config']['bytes_per_step_total'] == 146800640
    slowest = 1e-3 * (1.0 + 0.1 * (gpus - 1))
    assert line['value'] == pytest.approx(20 * 146800640 / slowest / 2**30, rel=1e-3)       # NOT multiplied by the rank count
    assert line['roofline']['algorithmic_bytes_per_launch'] == (n // gpus) * (2 * 2 + 3 / 8)


def test_the_parent_prints_a_childs_traceback(tmp_path):
    r = _run(tmp_path, 'raise', gpus=4)
    assert r.returncode != 0 and r.stdout.strip() == ''
    assert 'bench.py rank 2 raised' in r.stderr and 'ValueError: the reason rank 2 died' in r.stderr
    assert 'rank 2 failed' in r.stderr


def test_ranks_on_the_same_device_of_a_big_enough_box_fail(tmp_path):
    r = _run(tmp_path, 'same_device', gpus=4)          # 4 devices reported, all ranks claim device 0
    assert r.returncode != 0 and r.stdout.strip() == ''
    assert 'pairwise distinct' in r.stderr


def test_file_barrier_orders_phases(tmp_path):
    sys.path.insert(0, str(ROOT))
    import bench
    world, log, lock = 4, [], threading.Lock()

    def rank_main(r):
        s = bench.FileSync(tmp_path, r, world, timeout=20.0)
        time.sleep(0.01 * r)
        with lock:
            log.append(('before', r))
        s.barrier('a')
        with lock:
            log.append(('after', r))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert [p for p, _ in log[:world]] == ['before'] * world and [p for p, _ in log[world:]] == ['after'] * world
    # an abort flag releases a rank that waits for a peer that will never come
    s = bench.FileSync(tmp_path, 0, 2, timeout=20.0)
    (tmp_path / 'abort').touch()
    with pytest.raises(RuntimeError):
        s.barrier('b')


def test_a_torchrun_environment_is_not_mistaken_for_a_self_launch():
    """WORLD_SIZE in the environment means "a launcher started me": bench.py must not start children of its own then (it
    would need a GPU to go further, so only the decision is checked, through --help-free argument parsing of the source)."""
    src = (ROOT / 'bench.py').read_text()
    assert "if 'WORLD_SIZE' in os.environ:" in src and 'os.exec' not in src.replace("never exec'ed", '').replace('Never exec', '')
