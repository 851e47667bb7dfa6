#!/bin/bash
# round 3, call b: ablation A/B of the LUT forward and the backward; new bench.py (single, self-launched 2 ranks sharing the GPU,
# torchrun 2 ranks over gloo); RoBERTa module route vs the reference's raw-op route
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
S=scratch/libfewbit_hip
timeout 600 python scratch/ablate.py fwd base=${S}_sweep.so noact=${S}_lutA1.so nolookup=${S}_lutA2.so nobuild_nolookup=${S}_lutA6.so nostate=${S}_lutA8.so copyonly=${S}_lutA15.so base2=${S}_sweep.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03b_ablate_fwd.txt
timeout 600 python scratch/ablate.py bwd base=${S}_sweep.so nogather=${S}_bwdA1.so nostate=${S}_bwdA2.so copyonly=${S}_bwdA3.so plainstore=${S}_bwdA4.so base2=${S}_sweep.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03b_ablate_bwd.txt
SIZE=33554432 timeout 600 python scratch/ablate.py bwd base=${S}_sweep.so nogather=${S}_bwdA1.so nostate=${S}_bwdA2.so copyonly=${S}_bwdA3.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03b_ablate_bwd.txt
echo "== bench default-ish"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03b_bench_k20.json 2> gpurun_out/r03b_bench_k20.err; tail -c 600 gpurun_out/r03b_bench_k20.err; python -c "
import json; d=json.load(open('gpurun_out/r03b_bench_k20.json')); print({k:d[k] for k in ('value','ms_per_step','pct_of_hbm_roofline','pct_of_hbm_roofline_event_timed','fwd_us','bwd_us')}); print(d['roofline']); print(d['op_level']); print(d['cold']); print({k:v.get('cpu_baseline') for k,v in d['configs'].items()}); print(d['cpu_baseline'], d.get('cpu_baseline_1thread'))"
echo "== bench --gpus 2 self-launched"; timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r03b_bench_2self.json 2> gpurun_out/r03b_bench_2self.err; echo rc=$?; tail -c 400 gpurun_out/r03b_bench_2self.err; cut -c1-900 gpurun_out/r03b_bench_2self.json
echo "== bench torchrun 2 ranks"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 5 2> gpurun_out/r03b_bench_2torchrun.err | tail -1 > gpurun_out/r03b_bench_2torchrun.json; echo rc=$?; tail -c 300 gpurun_out/r03b_bench_2torchrun.err; cut -c1-600 gpurun_out/r03b_bench_2torchrun.json
echo "== roberta"; for dt in fp32 bf16; do timeout 900 python tools/roberta_bench.py --dtype $dt --route both 2> gpurun_out/r03b_roberta_$dt.err | tail -1 > gpurun_out/r03b_roberta_$dt.json; tail -c 300 gpurun_out/r03b_roberta_$dt.err; python -c "
import json; d=json.load(open('gpurun_out/r03b_roberta_$dt.json')); print('$dt', {k: (round(d[k]['ms_per_step'],2), round(d[k]['peak_bytes']/2**30,3), d[k]['saved_for_backward_bytes']) for k in ('vanilla','fewbit','op')}, d['op_vs_module'])"; done
