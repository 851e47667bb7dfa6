#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
for v in default bnt; do
  if [ $v = default ]; then unset FEWBIT_HIP_LIB; else export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_$v.so; fi
  SIZES=16777216,33554432,67108864 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids
done
unset FEWBIT_HIP_LIB
python tools/roberta_bench.py --dtype bf16 --steps 20 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('roberta bf16 default', d['vanilla']['ms_per_step'], d['fewbit']['ms_per_step'])"
cp fewbit_amd/libfewbit_hip.so /tmp/keep.so; cp scratch/libfewbit_hip_bnt.so fewbit_amd/libfewbit_hip.so
python tools/roberta_bench.py --dtype bf16 --steps 20 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('roberta bf16 bwd-nt ', d['vanilla']['ms_per_step'], d['fewbit']['ms_per_step'])"
cp /tmp/keep.so fewbit_amd/libfewbit_hip.so
python tools/roberta_bench.py --dtype bf16 --steps 20 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('roberta bf16 default', d['vanilla']['ms_per_step'], d['fewbit']['ms_per_step'])"
