#!/bin/bash
# round 3, call c: new GPU tests (multi-shard C4 digests, whole-view in-place), in-place vs out-of-place timing, bench line,
# rocprofv3 kernel trace of the RoBERTa step (in-situ durations of the few-bit kernels)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_ops.py -x -q 2>&1 | tail -4 | tee gpurun_out/r03c_tests.log
S=scratch/libfewbit_hip
for ip in 0 1; do for w in fwd bwd step; do INPLACE=$ip ROUNDS=5 timeout 300 python scratch/ablate.py $w base=${S}_sweep.so 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r03c_inplace.txt
SIZE=50331648 DT=f32 ROUNDS=3 timeout 300 python scratch/ablate.py step base=${S}_sweep.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03c_inplace.txt
SIZE=50331648 DT=f32 INPLACE=1 ROUNDS=3 timeout 300 python scratch/ablate.py step base=${S}_sweep.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03c_inplace.txt
echo "== bench k20"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03c_bench_k20.json 2> gpurun_out/r03c_bench_k20.err; python -c "
import json; d=json.load(open('gpurun_out/r03c_bench_k20.json')); print({k:d[k] for k in ('value','ms_per_step','pct_of_hbm_roofline','pct_of_hbm_roofline_event_timed','fwd_us','bwd_us')}); r=d['roofline']; print({k:r[k] for k in ('frac','frac_timed_region','steady_step_us','avg_launch_us','frac_cold')}); print(d['op_level'])"
echo "== in-situ"; mkdir -p gpurun_out/r03c_insitu
for dt in fp32 bf16; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03c_insitu/$dt -o rob -- python3 tools/roberta_bench.py --dtype $dt --only fewbit --steps 6 > gpurun_out/r03c_insitu/$dt.json 2> gpurun_out/r03c_insitu/$dt.err
  tail -c 300 gpurun_out/r03c_insitu/$dt.err
done
find gpurun_out/r03c_insitu -name "*kernel_stats.csv" | head; for f in $(find gpurun_out/r03c_insitu -name "*kernel_stats.csv"); do echo $f; grep -i "fewbit\|quantize_\|stepwise1" $f | cut -c1-260 | head -6; done
du -sh gpurun_out/r03c_insitu; find gpurun_out/r03c_insitu -name "*kernel_trace.csv" -size +20M -delete
