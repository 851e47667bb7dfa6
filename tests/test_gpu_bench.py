"""The driver-facing contract of bench.py on hardware: one JSON line with the fields the driver reads, for N = 1 and for the
self-launched N = 2 / 8 (ranks sharing this box's GPU when it has fewer devices).  The timed region is tiny (--steps 5); what
is checked is the shape of the line and the internal consistency of its numbers, not performance.  --scaling strong: the
ranks' result bytes (SHA-256 per rank) must be the slices of the unsharded result."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config')


def _bench_raw(*args, **extra_env):
    env = dict(os.environ, **extra_env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    return subprocess.run([sys.executable, str(ROOT / 'bench.py'), *args], env=env, capture_output=True, text=True, timeout=900)


def _bench(*args):
    r = _bench_raw(*args)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check_contract(line, n_gpus, steps, warmup, scaling='weak'):
    for key in CONTRACT:
        assert key in line, key
    assert line['n_gpus'] == n_gpus and line['steps'] == steps and line['warmup'] == warmup
    assert line['unit'] == 'GiB/s' and line['higher_is_better'] is True and line['scaling'] == scaling and line['vs_baseline'] is None
    assert line['dtype'] == 'bf16' and line['data'] == 'synthetic' and 'workload' in line['config'] and 'model' not in line['config']
    # value is the whole job: bytes of ALL ranks / the reported time per step
    total = line['config']['bytes_per_step_total']
    if scaling == 'weak':
        assert total == line['config']['bytes_per_step_per_gpu'] * n_gpus
    assert line['value'] == pytest.approx(total / (line['ms_per_step'] * 1e-3) / 2**30, rel=2e-3)
    # (ranks sharing ONE device also share its 256 MiB Infinity Cache: two warm 134 MiB working sets read x and gy from the
    # cache, so the algorithmic rate can exceed the HBM peak there -- seen: 107 %; never on a device of its own)
    assert 0.0 < line['pct_of_hbm_roofline'] < (100.0 if not line.get('shared_gpu') else 250.0)
    roof = line['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel'):
        assert key in roof, key
    assert roof['bound'] == 'hbm' and roof['peak'] == 8000.0 and roof['unit'] == 'GB/s'
    assert roof['frac'] == pytest.approx(roof['achieved'] / roof['peak'], abs=2e-4) and 0.0 < roof['frac'] < 1.0
    assert roof['kernel'].startswith('quantize_forward_lut_kernel<gelu, bf16, 3 bits')      # the dispatch's own answer


def test_single_gpu_line():
    line = _bench('--steps', '5', '--warmup', '2', '--no-extras', '--no-cpu-baseline')
    _check_contract(line, 1, 5, 2)
    assert 'shared_gpu' not in line


def test_pmc_traffic_is_measured_by_the_run_itself():
    """roofline.traffic: two rocprofv3 --pmc child runs of bench.py's own timed command (FETCH_SIZE x2 + WRITE_SIZE, KiB)."""
    import argparse
    import importlib.util
    import shutil
    if not (shutil.which('rocprofv3') or os.path.exists('/opt/rocm/bin/rocprofv3')):
        pytest.skip('rocprofv3 not on this box')
    spec = importlib.util.spec_from_file_location('bench_module', ROOT / 'bench.py')
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    if bench.under_profiler():
        pytest.skip('the test-suite itself runs under a profiler')
    algorithmic = bench.step_bytes(bench.CONFIGS['c2'])[1]
    assert algorithmic == 4096 * 4096 * (2 * 2 + 3 / 8)
    got = bench.measure_traffic(argparse.Namespace(config='c2'), algorithmic, steps=10)
    assert got['dispatches'] >= 10
    # nothing re-read, nothing written twice: x is served from the Infinity Cache or HBM (both counted at the fabric), y + state written once
    assert 0.98 <= got['over_algorithmic'] <= 1.03, got
    assert got['write_bytes'] >= 4096 * 4096 * (2 + 3 / 8)


def test_sketch_extra_quotes_the_reference_ratio_at_both_widths():
    """the `sketch` object of the default line: proj_dim_ratio 0.2 (p = 3276 of 16384 rows, the reference README's ratio) at 3072
    and 768 features, both distributions, each with its roofline, the torch pair timed beside it and a `wins_vs_torch` flag.
    STRUCTURE only: keys, roofline arithmetic, the plan, and that the `workload` text says where S lives exactly as the plan does
    (`s_fragment_bytes` == 0 <=> "never materialised").  The speed comparison itself is the line's `wins_vs_torch` flag (the
    Gaussian margin at 3072 features is 7-8 %, of the order of the pool's box-to-box scatter: no wall-clock assertion in a
    correctness suite beyond a perf smoke bar of 1.25x)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_module', ROOT / 'bench.py')
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import torch
    got = bench.measure_sketch(torch.device('cuda', torch.cuda.current_device()))
    assert set(got['ratio_0.2']) == {'16384x3072', '16384x768'}
    for shape, rec in got['ratio_0.2'].items():
        features = int(shape.split('x')[1])
        for dist, pair in (('rademacher', 'randint_plus_matmul_us'), ('gaussian', 'randn_plus_matmul_us')):
            e = rec[dist]
            flops = 2.0 * 3276 * 16384 * features
            assert e['roofline']['bound'] == 'mfma' and e['roofline']['peak'] == 2500.0
            assert abs(e['roofline']['achieved'] - flops / e['us'] / 1e6) <= 0.02 * e['roofline']['achieved'] and 0.05 < e['roofline']['frac'] < 1.0
            assert e['torch_us'] == rec['torch'][pair] and e['wins_vs_torch'] == (e['us'] <= e['torch_us'])
            assert e['plan']['grid'][0] == features // 256
            # where S lives: the text, the byte field and the plan agree
            assert e['s_fragment_bytes'] == e['plan']['s_fragment_bytes']
            assert ('never materialised' in e['workload']) == (e['plan']['s_fragment_bytes'] == 0), e['workload']
            assert ('from memory' in e['plan']['kernel']) == (e['plan']['s_fragment_bytes'] > 0)
            if e['plan']['s_fragment_bytes']:
                assert str(e['plan']['s_fragment_bytes']) in e['workload'] and e['workspace_bytes'] >= e['s_fragment_bytes']
        assert rec['gaussian']['plan']['s_fragment_bytes'] > 0 and rec['rademacher']['plan']['s_fragment_bytes'] == 0
        for dist, e in rec['fp32_input'].items():                   # fp32 input: labelled as bf16-operand arithmetic wherever it is quoted
            assert 'bf16 operands' in e['operands'] and 'bf16 operands' in e['workload'], e
    assert got['wins_vs_torch'] == all(rec[d]['wins_vs_torch'] for rec in got['ratio_0.2'].values() for d in ('rademacher', 'gaussian'))
    for rec in got['ratio_0.2'].values():                           # perf smoke only (a broken launch policy, not a 5 % drift)
        assert rec['rademacher']['us'] <= 1.25 * rec['rademacher']['torch_us'] and rec['gaussian']['us'] <= 1.25 * rec['gaussian']['torch_us'], rec
    # the reference's sampled transforms stand beside the dense sketches, each against its byte floor
    assert set(got['sampled_transform']) == {'16384x768_bf16', '16384x768_fp32', '16384x3072_bf16', '16384x3072_fp32'}
    for name, rec in got['sampled_transform'].items():
        assert 'error' not in rec, rec
        features, es = int(name.split('x')[1].split('_')[0]), 2 if name.endswith('bf16') else 4
        assert rec['byte_floor']['bytes'] == (16384 + 3276) * features * es
        for kind in ('dct', 'dft'):
            assert rec[kind]['us'] > rec['byte_floor']['us_at_8TBs'] and rec[kind]['path']


def test_self_launched_ranks_line():
    line = _bench('--gpus', '2', '--steps', '5', '--warmup', '2')
    _check_contract(line, 2, 5, 2)
    assert len(line['per_gpu_us_per_step']) == 2
    import torch
    if torch.cuda.device_count() < 2:
        assert line.get('shared_gpu') is True
    assert 'started 2 child processes itself' in line['config']['launcher']


def test_eight_self_launched_ranks_line():
    """the launch path the 8-GPU lease will take (python bench.py --gpus 8): 8 children, file barriers, one line"""
    line = _bench('--gpus', '8', '--steps', '5', '--warmup', '2')
    _check_contract(line, 8, 5, 2)
    assert len(line['per_gpu_us_per_step']) == 8 and len(line['per_gpu_device']) == 8
    import torch
    ndev = torch.cuda.device_count()
    if ndev >= 8:
        assert sorted(line['per_gpu_device']) == list(range(8)) and 'shared_gpu' not in line
    else:
        assert line.get('shared_gpu') is True and set(line['per_gpu_device']) <= set(range(ndev))


def test_a_failing_rank_shows_its_traceback():
    r = _bench_raw('--gpus', '2', '--steps', '5', '--warmup', '2', '--launch-timeout', '300', FEWBIT_BENCH_INJECT_FAILURE='1')
    assert r.returncode != 0 and r.stdout.strip() == ''
    assert 'bench.py rank 1 raised' in r.stderr and 'RuntimeError: injected failure on rank 1' in r.stderr


def test_emulated_world_line_projects_from_one_slice():
    line = _bench('--gpus', '1', '--scaling', 'strong', '--emulate-world', '8', '--steps', '20', '--warmup', '5', '--no-extras', '--no-cpu-baseline')
    assert line['n_gpus'] == 1 and line['emulated_world'] == 8 and line['config']['elements_per_gpu'] == 4096 * 4096 // 8
    assert line['projected']['n_gpus'] == 8 and line['projected']['value_GiB_s'] == pytest.approx(8 * line['value'], rel=1e-3)
    assert 0 < line['projected']['pct_of_hbm_roofline'] < 100


@pytest.mark.parametrize('gpus', (1, 2))
def test_strong_scaling_bytes_are_slices_of_the_unsharded_result(gpus):
    """--scaling strong --digests: rank r's (y, state, gx) are bit for bit the slices [begin_r, end_r) of what ONE launch over
    the whole seeded 4096x4096 tensor gives (computed here, in this process, through the same C-ABI)."""
    import hashlib
    import torch
    from fewbit_amd import cabi
    from fewbit_amd.sharding import shard_range, state_range
    import bench
    line = _bench('--gpus', str(gpus), '--steps', '5', '--warmup', '2', '--scaling', 'strong', '--digests', '--no-extras', '--no-cpu-baseline')
    _check_contract(line, gpus, 5, 2, scaling='strong')
    cfg = bench.CONFIGS['c2']
    n = cfg['rows'] * cfg['cols']
    assert line['config']['elements_total'] == n and sum(s[1] - s[0] for s in line['per_gpu_span']) == n
    x = torch.randn(n, generator=torch.Generator().manual_seed(0)).to(torch.bfloat16).cuda()
    gy = torch.randn(n, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).cuda()
    borders, levels = bench.load_tables(cfg, 'cuda')
    y, state = cabi.quantize_forward('gelu', x, borders)
    gx = cabi.quantize_backward(gy, state, levels)

    def sha(t):
        return hashlib.sha256(t.cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()

    for r in range(gpus):
        b, e = shard_range(n, gpus, r)
        assert line['per_gpu_span'][r] == [b, e]
        sb, se = state_range(b, e, cfg['bits'])
        got = line['per_gpu_sha256'][r]
        assert got['y'] == sha(y[b:e]) and got['gx'] == sha(gx[b:e]) and got['state'] == sha(state[sb:se]), f'rank {r}'


def test_the_drivers_torchrun_form_with_two_ranks():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` (the driver's N > 1 command): gloo carries the
    barrier and the max only; one line from rank 0; --scaling strong --digests there too"""
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1']
    for port, extra, scaling in (('29531', (), 'weak'), ('29532', ('--scaling', 'strong', '--digests'), 'strong')):
        r = subprocess.run(base + ['--master-port', port, str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2', *extra],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1, r.stdout[-2000:]
        line = json.loads(lines[0])
        _check_contract(line, 2, 5, 2, scaling=scaling)
        assert 'torch.distributed.run' in line['config']['launcher'] and len(line['per_gpu_device']) == 2
        if scaling == 'strong':
            assert sum(line['per_gpu_elements']) == 4096 * 4096 and len(line['per_gpu_sha256']) == 2
