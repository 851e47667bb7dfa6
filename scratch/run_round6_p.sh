#!/bin/bash
# round 6, call p: the samples sorted by residue class in pass A (sort_rows): whole GPU suite, DCT soak, times incl. large p and explicit idx
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -3 | cut -c1-300
FEWBIT_DCT_FUZZ_CASES=1500 timeout 900 python3 -m pytest tests/test_gpu_dct.py -q -k fuzz 2>&1 | tail -2
OUT=gpurun_out/r06p_dct_times.txt; : > $OUT
for shape in "16384 768 3276 bf16" "16384 3072 3276 bf16" "16384 768 16384 bf16" "65536 768 13107 bf16" "65536 64 65536 bf16" "4096 768 819 bf16" "12288 768 2457 bf16" "256 768 51 bf16"; do
  for mode in explicit seeded; do
    timeout 120 python3 tools/dct_run.py $shape 200 40 $mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$shape', '$mode', d['event_us_per_call'])" >> $OUT
  done
done
cat $OUT
