"""The driver-facing contract of bench.py on hardware: one JSON line with the fields the driver reads, for N = 1 and for the
self-launched N = 2 (two ranks sharing this box's GPU).  The timed region is tiny (--steps 5); what is checked is the shape
of the line and the internal consistency of its numbers, not performance."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config')


def _bench(*args):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), *args], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check_contract(line, n_gpus, steps, warmup):
    for key in CONTRACT:
        assert key in line, key
    assert line['n_gpus'] == n_gpus and line['steps'] == steps and line['warmup'] == warmup
    assert line['unit'] == 'GiB/s' and line['higher_is_better'] is True and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert line['dtype'] == 'bf16' and line['data'] == 'synthetic' and 'workload' in line['config'] and 'model' not in line['config']
    # value is the whole job: bytes of ALL ranks / the reported time per step
    total = line['config']['bytes_per_step_per_gpu'] * n_gpus
    assert line['value'] == pytest.approx(total / (line['ms_per_step'] * 1e-3) / 2**30, rel=2e-3)
    assert 0.0 < line['pct_of_hbm_roofline'] < 100.0
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


def test_self_launched_ranks_line():
    line = _bench('--gpus', '2', '--steps', '5', '--warmup', '2')
    _check_contract(line, 2, 5, 2)
    assert len(line['per_gpu_us_per_step']) == 2
    import torch
    if torch.cuda.device_count() < 2:
        assert line.get('shared_gpu') is True
    assert 'started 2 child processes itself' in line['config']['launcher']
