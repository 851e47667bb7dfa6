#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_bench.py -x -q -k "capture or sketch_extra" 2>&1 | tail -5 | tee gpurun_out/r05w_tests.log
bash scratch/run_round5_q.sh
