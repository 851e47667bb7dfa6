#!/bin/bash
# round 5, call p: the whole GPU suite on the final tree, then the soaks: 3000-case sketch fuzz, 1500-seed hot-path fuzz
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r05p_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r05p_smoke.log
FEWBIT_SKETCH_FUZZ_CASES=3000 timeout 1500 python -m pytest tests/test_gpu_sketch.py -q -m gpu -k fuzz 2>&1 | tail -3 | tee gpurun_out/r05p_sketch_soak_fuzz.txt
FEWBIT_FUZZ_SEEDS=1500 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 | tee gpurun_out/r05p_soak_fuzz.txt
