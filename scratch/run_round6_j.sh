#!/bin/bash
# round 6, call j: TIMING experiment: both DCT passes in ONE launch (no dependency tracking: pass B reads the previous call's intermediate), pass B `lag` tiles behind pass A
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r06j_dct_fused.txt; : > $OUT
export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_dct_fused.so
for shape in "16384 768 3276 bf16" "16384 3072 3276 bf16" "16384 768 3276 f32"; do
  for lag in -1 99 1 2 3 4 6 -1; do
    export FB_DCT_LAG=$lag
    timeout 120 python3 tools/dct_run.py $shape 200 40 ${MODE:-explicit} 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$shape', 'lag $lag', d['event_us_per_call'])" >> $OUT
  done
done
cat $OUT
