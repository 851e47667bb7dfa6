#!/bin/bash
# round 5, call y: the opt-in rows of the RoBERTa table on the final tree: sketch_dtype=bf16 on the fp32 model (bf16 projection, bf16 final GEMM)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "fp32 gaussian" "fp32 rademacher"; do set -- $v
    timeout 600 python3 tools/roberta_bench.py --table --dtype $1 --matmul $2 --steps 6 --sketch-bf16 2>> gpurun_out/r05y_roberta.err | tail -1 > gpurun_out/r05y_roberta_table_$1_$2_sketchbf16.json
done
bash scratch/run_round5_q.sh
