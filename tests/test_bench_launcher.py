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
    ap.add_argument('--config'); ap.add_argument('--settle-ms'); ap.add_argument('--sync-dir')
    args = ap.parse_args()
    mode = os.environ.get('STUB_MODE', 'ok')
    rank, world = args.worker_rank, args.gpus
    if mode == 'die_early' and rank == 1:
        sys.exit(3)
    sync = bench.FileSync(args.sync_dir, rank, world, timeout=30.0)
    sync.barrier('start')
    wall = 1e-3 * (1.0 + 0.1 * rank)             # rank r "takes" (1 + 0.1 r) ms for the K steps
    sync.barrier('stop')
    if mode == 'die_late' and rank == world - 1:
        sys.exit(4)
    res = dict(rank=rank, device=rank, n_devices=world, wall_s=wall, event_s=0.9 * wall, own_wall_s=wall, own_event_s=0.9 * wall,
               settle_steps=0, set_bytes=140509184)
    if rank == 0:
        res.update(fwd_us=12.0, bwd_us=11.0, steady_step_us=23.5,
                   kernels=[dict(kernel='stub_fwd', blocks=1, threads=1, blocks_per_cu=1, chunk=0, u=1, bits=3),
                            dict(kernel='stub_bwd', blocks=1, threads=1, blocks_per_cu=1, chunk=0, u=1, bits=3)])
    tmp = Path(args.sync_dir) / f'result.{{rank}}.json.tmp'
    tmp.write_text(json.dumps(res))
    tmp.rename(Path(args.sync_dir) / f'result.{{rank}}.json')
''')


def _run(tmp_path, mode, gpus=4, steps=20):
    stub = tmp_path / 'stub_worker.py'
    stub.write_text(STUB.format(root=str(ROOT)))
    env = dict(os.environ, FEWBIT_BENCH_WORKER=str(stub), STUB_MODE=mode)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    return subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', str(gpus), '--steps', str(steps), '--warmup', '5',
                           '--launch-timeout', '60'], env=env, capture_output=True, text=True, timeout=120)


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
