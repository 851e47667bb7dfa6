#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r02n_pytest.log 2>&1
tail -4 gpurun_out/r02n_pytest.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['traffic_source'][:40], d['cold']['frac'])"
