#!/bin/bash
# round 5, call o: final evidence on the final build: the sketch table, the RoBERTa A/B (both dtypes) and table rows, the default bench line
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scratch/sketch_bench.py > gpurun_out/r05o_sketch_bench.log 2>&1; cp gpurun_out/sketch_bench.json gpurun_out/r05o_sketch_bench.json
for dt in fp32 bf16; do timeout 900 python scratch/roberta_ab.py $dt 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05o_roberta_ab_$dt.txt; done
for v in "fp32 gaussian" "fp32 rademacher" "bf16 gaussian" "bf16 rademacher"; do set -- $v
    timeout 600 python3 tools/roberta_bench.py --table --dtype $1 --matmul $2 --steps 6 2>> gpurun_out/r05o_roberta.err | tail -1 > gpurun_out/r05o_roberta_table_$1_$2.json
done
timeout 600 python bench.py > gpurun_out/r05o_bench_line.json 2> gpurun_out/r05o_bench.err
bash tools/profile_insitu_sketch.sh r05o > gpurun_out/r05o_insitu_sketch.log 2>&1
