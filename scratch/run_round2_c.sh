#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r02c_headvar.log
for v in default ea e0; do
  if [ $v = default ]; then unset FEWBIT_HIP_LIB; else export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_$v.so; fi
  python scratch/headvar.py 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02c_headvar.log
done
unset FEWBIT_HIP_LIB
cat gpurun_out/r02c_headvar.log
python scratch/timeline.py 0.0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02c_timeline.log
python scratch/timeline.py 2.0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02c_timeline.log
for a in "20 5" "200 5" "200 20" "2000 50"; do set -- $a; python bench.py --steps $1 --warmup $2 --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench K=$1 W=$2', d['value'], d['ms_per_step'], d['timing']['wall_ms_per_step'], d['fwd_us'], d['bwd_us'])"; done
python -m pytest tests/test_gpu_parity.py -q -k "fp32_forward_in_raw or every_16bit or digests" 2>&1 | tail -3
FEWBIT_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 100 --warmup 10 2>&1 | tail -2 | cut -c1-600
