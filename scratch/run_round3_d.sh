#!/bin/bash
# round 3, call d: in-place stores plain vs nontemporal (run-time flag), parity subset with the new build, bench wall overhead,
# tiny-tensor floor (stream bench at 4 MiB)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
S=scratch/libfewbit_hip_sweep.so
for w in fwd bwd step; do INPLACE=1 ROUNDS=5 timeout 300 python scratch/ablate.py $w plain=$S@nt_inplace=0 nontemporal=$S@nt_inplace=1 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03d_inplace_stores.txt
for w in fwd bwd step; do INPLACE=0 ROUNDS=5 timeout 300 python scratch/ablate.py $w out_of_place=$S 2>&1 | grep -v amdgpu.ids; done | tee -a gpurun_out/r03d_inplace_stores.txt
SIZE=50331648 DT=f32 INPLACE=1 ROUNDS=3 timeout 300 python scratch/ablate.py step plain=$S@nt_inplace=0 nontemporal=$S@nt_inplace=1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03d_inplace_stores.txt
SIZE=50331648 DT=f32 INPLACE=0 ROUNDS=3 timeout 300 python scratch/ablate.py step out_of_place=$S 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03d_inplace_stores.txt
SIZE=67108864 DT=f16 FN=silu K=4 INPLACE=1 ROUNDS=3 timeout 300 python scratch/ablate.py step plain=$S@nt_inplace=0 nontemporal=$S@nt_inplace=1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03d_inplace_stores.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 | tee gpurun_out/r03d_tests.log
echo "== bench k20"; for i in 1 2; do timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r03d_bench_k20_$i.json 2> gpurun_out/r03d_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03d_bench_k20_$i.json')); print({k:d[k] for k in ('value','ms_per_step','pct_of_hbm_roofline','pct_of_hbm_roofline_event_timed','fwd_us','bwd_us')}, d['timing']['event_ms_per_step'])"; done
timeout 900 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r03d_bench_default.json 2>> gpurun_out/r03d_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03d_bench_default.json')); print({k:d[k] for k in ('value','ms_per_step','pct_of_hbm_roofline','pct_of_hbm_roofline_event_timed','fwd_us','bwd_us')}, d['timing']['event_ms_per_step'], d['roofline']['frac'])"
echo "== tiny"; timeout 300 scratch/stream_bench 4 > gpurun_out/r03d_stream_4MiB.txt 2>&1; cat gpurun_out/r03d_stream_4MiB.txt
