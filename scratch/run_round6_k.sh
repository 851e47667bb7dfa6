#!/bin/bash
# round 6, call k: experiment: the DCT intermediate in bf16 for bf16 input (half the intermediate's bytes): time and error against float64
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r06k_dct_inter16.txt; : > $OUT
for v in ${VARIANTS:-prod inter16}; do
  if [ $v = prod ]; then unset FEWBIT_HIP_LIB; else export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_dct_$v.so; fi
  echo "== $v" >> $OUT
  python3 scratch/dct_accuracy.py 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
VARIANTS="${VARIANTS:-prod inter16} prod" bash scratch/run_round6_d.sh | grep -v "f32"
