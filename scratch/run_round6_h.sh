#!/bin/bash
# round 6, call h: the sampled rows as a function of the seed (fewbit_hip_sampled_dct_seeded): tests, seeded against explicit idx per shape, the RoBERTa rows
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_dct.py tests/test_gpu_linear.py tests/test_gpu_sketch.py -m gpu -q -x 2>&1 | tail -8 | cut -c1-300
OUT=gpurun_out/r06h_dct_seeded.txt; : > $OUT
for shape in "16384 768 3276 bf16" "16384 3072 3276 bf16" "16384 768 3276 f32" "16384 3072 3276 f32" "65536 768 13107 bf16" "4096 768 819 bf16" "16384 768 16384 bf16"; do
  for mode in explicit seeded explicit seeded; do
    timeout 120 python3 tools/dct_run.py $shape 200 40 $mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$shape', '$mode', d['event_us_per_call'])" >> $OUT
  done
done
cat $OUT
bash tools/profile_dct.sh r06h 16384 768 3276 bf16 seeded > gpurun_out/r06h_prof_seeded.log 2>&1; grep -E "pass_[ab]|sum of" gpurun_out/r06h_prof_seeded.log | cut -c1-200
bash tools/profile_dct.sh r06h 16384 768 3276 bf16 explicit > gpurun_out/r06h_prof_explicit.log 2>&1; grep -E "pass_[ab]|sum of" gpurun_out/r06h_prof_explicit.log | cut -c1-200
for dt in bf16 fp32; do
  timeout 600 python3 tools/roberta_bench.py --table --dtype $dt --matmul dct --steps 6 2>/dev/null | tail -1 > gpurun_out/r06h_roberta_table_${dt}_dct.json
  python3 - <<PY
import json
d=json.load(open('gpurun_out/r06h_roberta_table_${dt}_dct.json'))
print('${dt}', json.dumps(d)[:1500])
PY
done
