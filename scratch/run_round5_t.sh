#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
FEWBIT_SKETCH_FUZZ_CASES=3000 timeout 1500 python -m pytest tests/test_gpu_sketch.py -q -m gpu -k fuzz 2>&1 | tail -3 | tee gpurun_out/r05t_sketch_soak_fuzz.txt
timeout 1500 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_linear.py -q -m gpu 2>&1 | tail -2 | tee gpurun_out/r05t_tests.log
