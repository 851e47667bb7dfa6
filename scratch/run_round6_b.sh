#!/bin/bash
# round 6, call b: the tree with the rejected experiments stripped and the C-ABI frozen (version 5, version script): the whole GPU suite, the default
# bench line (headline within +-1 % of BENCH_r05?), then both fuzz soaks
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06b_pytest_gpu.txt; cat gpurun_out/r06b_pytest_gpu.txt | tail -4
timeout 900 python3 bench.py > gpurun_out/r06b_bench_line.json 2> gpurun_out/r06b_bench.err; cut -c1-700 gpurun_out/r06b_bench_line.json
FEWBIT_SKETCH_FUZZ_CASES=3000 timeout 1500 python3 -m pytest tests/test_gpu_sketch.py -q -m gpu -k fuzz 2>&1 | tail -3 | tee gpurun_out/r06b_sketch_soak_fuzz.txt
FEWBIT_FUZZ_SEEDS=1500 timeout 3000 python3 -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 | tee gpurun_out/r06b_soak_fuzz.txt
