#!/bin/bash
# round 5, call v: final tree (woven generator in the fused 256 x 256 Gaussian kernel): whole GPU suite, smoke, sketch table, RoBERTa A/B, bench line
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r05v_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r05v_smoke.log
timeout 900 python scratch/sketch_bench.py > gpurun_out/r05v_sketch_bench.log 2>&1; cp gpurun_out/sketch_bench.json gpurun_out/r05v_sketch_bench.json
for dt in fp32 bf16; do timeout 900 python scratch/roberta_ab.py $dt 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05v_roberta_ab_$dt.txt; done
timeout 600 python3 tools/roberta_bench.py --table --dtype fp32 --matmul gaussian --steps 6 2>> gpurun_out/r05v_roberta.err | tail -1 > gpurun_out/r05v_roberta_table_fp32_gaussian.json
timeout 600 python bench.py > gpurun_out/r05v_bench_line.json 2> gpurun_out/r05v_bench.err
