#!/bin/bash
# round 5, call d: sketch + layer tests on the fixed plan; which settings broke the fuzz case; in-situ PMC session of the in-place forward;
# the sketch table (scratch/sketch_bench.py) and the RoBERTa-base table rows on the new Gaussian path
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python scratch/dbg_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/r05d_dbg_fuzz.txt
timeout 1500 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_linear.py tests/test_portable_build.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r05d_tests.log
timeout 1500 bash tools/profile_insitu_pmc.sh r05 > gpurun_out/r05d_insitu_pmc.log 2>&1
timeout 900 python scratch/sketch_bench.py > gpurun_out/r05d_sketch_bench.log 2>&1; cp gpurun_out/sketch_bench.json gpurun_out/r05d_sketch_bench.json
for v in "fp32 gaussian" "fp32 rademacher" "bf16 gaussian" "bf16 rademacher"; do
    set -- $v
    timeout 600 python3 tools/roberta_bench.py --table --dtype $1 --matmul $2 --steps 6 2>> gpurun_out/r05d_roberta.err | tail -1 > gpurun_out/r05d_roberta_table_$1_$2.json
done
tail -3 gpurun_out/r05d_sketch_bench.log | cut -c1-600
