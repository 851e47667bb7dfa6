#!/bin/bash
# round 3, call s: long soak -- 1500-seed differential fuzz (launch shape randomised per case), full GPU suite twice
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
FEWBIT_FUZZ_SEEDS=1500 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 | tee gpurun_out/r03s_soak_fuzz_1500.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee gpurun_out/r03s_tests.log
timeout 600 python bench.py --steps 100000 --warmup 50 --no-extras --no-cpu-baseline > gpurun_out/r03s_bench_100k.json 2> gpurun_out/r03s_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03s_bench_100k.json')); print({k:d[k] for k in ('value','ms_per_step','pct_of_hbm_roofline','steps')})"
