#!/bin/bash
# second GPU pass: full GPU suite with the early-prefetch pipeline, variants (tile depth), host cost
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r02b_pytest.log 2>&1
tail -15 gpurun_out/r02b_pytest.log
rm -f gpurun_out/r02b_headvar.log
for v in default e0 u2 u4 b1 b4; do
  if [ $v = default ]; then unset FEWBIT_HIP_LIB; else export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_$v.so; fi
  python scratch/headvar.py 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02b_headvar.log
done
export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_u2.so; FEWBIT_HIP_LUT_BLOCKS_PER_CU=1 TAGX=+1bpc python scratch/headvar.py 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02b_headvar.log
export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_u4.so; FEWBIT_HIP_LUT_BLOCKS_PER_CU=1 TAGX=+1bpc python scratch/headvar.py 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02b_headvar.log
unset FEWBIT_HIP_LIB
python scratch/headvar.py 1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02b_headvar.log
cat gpurun_out/r02b_headvar.log
python scratch/hostcost.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02b_hostcost.log
for i in 1 2 3; do python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['fwd_us'], d['bwd_us'])"; done
