#!/bin/bash
# round 5, call af: the stand-alone sketch table and the bench line under the width rule of the policy for fp32 input
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
python3 scratch/sketch_bench.py > gpurun_out/r05af_sketch_bench.log 2>&1; tail -3 gpurun_out/r05af_sketch_bench.log
python3 bench.py > gpurun_out/r05af_bench_line.json 2> gpurun_out/r05af_bench.err; cut -c1-600 gpurun_out/r05af_bench_line.json
