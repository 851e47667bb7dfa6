#!/bin/bash
# round 6, call a: (1) the sketch's roofline settled under rocprofv3 (tools/profile_sketch.sh), (2) tools/sketch_bench.py settled the same way, with the
# reference's dct / dft estimators beside the dense sketches, (3) the RoBERTa-base table with --matmul dct / dft beside gaussian / rademacher on the same lease
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/profile_sketch.sh r06 > gpurun_out/r06a_profile_sketch.log 2>&1
timeout 1200 python3 tools/sketch_bench.py > gpurun_out/r06a_sketch_bench.log 2>&1; cp gpurun_out/sketch_bench.json gpurun_out/r06a_sketch_bench.json
for dt in bf16 fp32; do
  for mm in dct dft rademacher gaussian; do
    timeout 600 python3 tools/roberta_bench.py --table --dtype $dt --matmul $mm 2>gpurun_out/r06a_roberta_${dt}_$mm.err | tail -1 > gpurun_out/r06a_roberta_table_${dt}_$mm.json
  done
done
tail -5 gpurun_out/r06a_profile_sketch.log; tail -2 gpurun_out/r06a_sketch_bench.log | cut -c1-1500; for f in gpurun_out/r06a_roberta_table_*.json; do echo $f; python3 -c "
import json,sys
d=json.load(open('$f'))
print([(r['gelu'],r['linear'],r['ms_per_step'],r['step_time_ratio'],r['saving_pct']) for r in d['rows']])" 2>&1 | tail -1; done
