#!/bin/bash
# round 3, call i: the split (six translation units) build -- whole GPU suite, bench lines
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r03i_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03i_bench_k20.json 2> gpurun_out/r03i_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03i_bench_k20.json')); print({k:d[k] for k in ('value','ms_per_step','pct_of_hbm_roofline','fwd_us','bwd_us')}, d['roofline']['frac'], d['roofline']['kernel'])"
