#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r02m_pytest.log 2>&1
tail -4 gpurun_out/r02m_pytest.log
bash tools/profile_round.sh r02 2>&1 | tail -14
python bench.py --config c4 --no-cpu-baseline > gpurun_out/profiles_r02/r02_bench_line_c4.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/profiles_r02/r02_bench_line_k20.json 2>/dev/null
scratch/stream_bench 32 > gpurun_out/profiles_r02/r02_stream_bench_32MiB.txt 2>&1
scratch/stream_bench 128 > gpurun_out/profiles_r02/r02_stream_bench_128MiB.txt 2>&1
python scratch/hostcost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02m_hostcost.log; cp gpurun_out/hostcost.json gpurun_out/profiles_r02/r02_hostcost.json
python scratch/timeline.py 0.0 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_r02/r02_clock_transient_timeline.txt
cp gpurun_out/fp32_ulp.json gpurun_out/profiles_r02/r02_fp32_ulp.json
FEWBIT_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 200 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r02m_bench_2ranks_gloo_shared_gpu.json
python -c "
import json
for f in ('r02_bench_line','r02_bench_line_c4','r02_bench_line_k20'):
    d=json.load(open('gpurun_out/profiles_r02/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['pct_of_hbm_roofline'], d['roofline']['frac'], d.get('cold',{}).get('frac'))
d=json.load(open('gpurun_out/r02m_bench_2ranks_gloo_shared_gpu.json')); print('2 ranks sharing one GPU (gloo):', d['value'], d['ms_per_step'], d['n_gpus'])
"
