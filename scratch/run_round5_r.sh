#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
FEWBIT_SKETCH_FUZZ_CASES=3000 timeout 1500 python -m pytest tests/test_gpu_sketch.py -q -m gpu -k fuzz -x 2>&1 | grep -v "^E   .*tensor(\|^E    +" | tail -60 | cut -c1-900 > gpurun_out/r05r_sketch_soak_fuzz.txt
bash scratch/run_round5_q.sh
