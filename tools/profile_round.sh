#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box and summarise it into gpurun_out/profiles_<round>/ (copied into
# profiles/ after review).
#   usage (through gpurun):  bash tools/profile_round.sh r02
# Separate passes, as MI355X_MICROARCH.md prescribes (PMC never combined with other trace domains than --kernel-trace):
#   1. python3 bench.py                                   -> <round>_bench_line.json  (the driver-visible line, all extras)
#   2. rocprofv3 --kernel-trace --stats   bench.py --no-extras   -> kernel stats of the timed region's command (C2)
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE  bench.py --no-extras --steps 50   -> HBM traffic per launch (C2)
#   4. rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE   tools/config_runs.py
#      -> per-dispatch durations and traffic of EVERY config, warm and cold (split by tools/summarize_profiles.py)
#   5. bench.py --config c4 / --steps 20 / 2 ranks on the one GPU (self-launched and under torch.distributed.run), stream
#      micro-benchmark, host cost per call, clock-transient timeline, the fp32 ULP histogram written by the GPU test-suite
#   6. RoBERTa-base step (both routes) and the rocprofv3 kernel stats of the few-bit kernels inside it
set -u
R=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export TMPDIR=/tmp
RAW=$ROOT/gpurun_out/prof_$R
OUT=$ROOT/gpurun_out/profiles_$R
rm -rf "$RAW"; mkdir -p "$RAW" "$OUT"
python3 bench.py > "$OUT/${R}_bench_line.json" 2> "$RAW/bench.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/trace" -o bench -- python3 bench.py --no-extras --no-cpu-baseline > "$RAW/bench_under_trace.json" 2> "$RAW/trace.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$RAW/fetch" -o bench -- python3 bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2> "$RAW/fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$RAW/write" -o bench -- python3 bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2> "$RAW/write.err"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$RAW/cfg_trace" -o cfg -- python3 tools/config_runs.py "$RAW/cfg_trace/phases.json" 120 > /dev/null 2> "$RAW/cfg_trace.err"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$RAW/cfg_fetch" -o cfg -- python3 tools/config_runs.py "$RAW/cfg_fetch/phases.json" 30 > /dev/null 2> "$RAW/cfg_fetch.err"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$RAW/cfg_write" -o cfg -- python3 tools/config_runs.py "$RAW/cfg_write/phases.json" 30 > /dev/null 2> "$RAW/cfg_write.err"
python3 tools/summarize_profiles.py "$R" "$RAW" "$OUT"
# 5. the other driver-visible lines and the side measurements DESIGN.md and EXPERIMENTS.md quote
python3 bench.py --config c4 --no-cpu-baseline > "$OUT/${R}_bench_line_c4.json" 2>> "$RAW/bench.err"
python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$OUT/${R}_bench_line_k20.json" 2>> "$RAW/bench.err"
# N > 1 on the one GPU of this box (ranks SHARE it: validation of the launch paths, not a scaling point): bench.py starting
# its own children, and the driver's torch.distributed.run form (gloo process group for the barrier and the max only)
python3 bench.py --gpus 2 --steps 20 --warmup 5 > "$OUT/${R}_bench_line_2ranks_selflaunch_one_gpu.json" 2>> "$RAW/bench.err"
# the launch path an 8-GPU lease takes (8 / 4 self-launched ranks, here sharing the one GPU), weak (the driver's contract) ...
for spec in "8 c2" "4 c2" "8 c4"; do set -- $spec
    python3 bench.py --gpus $1 --config $2 --steps 20 --warmup 5 > "$OUT/${R}_bench_line_${1}ranks_${2}_shared.json" 2>> "$RAW/bench.err"
done
# ... and strong: ONE 4096x4096 (and one 16384x4096) tensor cut over N ranks; ranks sharing the GPU, and rank 0's slice of an
# M-way cut ALONE on the GPU (--emulate-world: what each of M separate GPUs runs; the line carries the projected M-GPU figure)
for n in 1 2 4 8; do
    python3 bench.py --gpus $n --scaling strong --steps 200 --warmup 20 --no-extras --no-cpu-baseline > "$OUT/${R}_strong_c2_${n}ranks_one_gpu.json" 2>> "$RAW/bench.err"
    python3 bench.py --gpus $n --config c4_tensor --scaling strong --steps 100 --warmup 20 --no-extras --no-cpu-baseline > "$OUT/${R}_strong_c4tensor_${n}ranks_one_gpu.json" 2>> "$RAW/bench.err"
done
for m in 2 4 8; do
    python3 bench.py --gpus 1 --scaling strong --emulate-world $m --steps 400 --warmup 50 --no-extras --no-cpu-baseline > "$OUT/${R}_strong_c2_emulated_world_$m.json" 2>> "$RAW/bench.err"
    python3 bench.py --gpus 1 --config c4_tensor --scaling strong --emulate-world $m --steps 200 --warmup 50 --no-extras --no-cpu-baseline > "$OUT/${R}_strong_c4tensor_emulated_world_$m.json" 2>> "$RAW/bench.err"
done
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
    bench.py --gpus 2 --steps 200 --warmup 10 2>> "$RAW/bench.err" | tail -1 > "$OUT/${R}_bench_line_2ranks_torchrun_one_gpu.json"
# 6. BASELINE config 5: RoBERTa-base step, module route and the reference's raw-operator route, and the few-bit kernels' own
#    durations INSIDE that step (rocprofv3 kernel trace; fp32 and bf16, both routes)
for dt in fp32 bf16; do
    python3 tools/roberta_bench.py --dtype $dt --route both 2>> "$RAW/roberta.err" | tail -1 > "$OUT/${R}_roberta_base_$dt.json"
    for route in fewbit op; do
        timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/insitu_${dt}_${route}" -o rob -- \
            python3 tools/roberta_bench.py --dtype $dt --only $route --steps 6 > /dev/null 2>> "$RAW/roberta.err"
        find "$RAW/insitu_${dt}_${route}" -name "*kernel_trace.csv" -delete
    done
done
python3 tools/summarize_profiles.py "$R" "$RAW" "$OUT" insitu > "$RAW/insitu.log" 2>&1
if [ -x scratch/bin/stream_bench ]; then
    scratch/bin/stream_bench 32 > "$OUT/${R}_stream_bench_32MiB.txt" 2>&1
    scratch/bin/stream_bench 128 > "$OUT/${R}_stream_bench_128MiB.txt" 2>&1
    scratch/bin/stream_bench 4 > "$OUT/${R}_stream_bench_4MiB.txt" 2>&1
fi
python3 scratch/hostcost.py > "$RAW/hostcost.log" 2>&1 && cp gpurun_out/hostcost.json "$OUT/${R}_hostcost.json"
# 7. the random-projection kernel: TFLOP/s per shape against torch.randn/randint + torch.matmul, the README
#    table of RoBERTa-base with the native and the torch sketches, rocprofv3 kernel stats + PMC of one shape
python3 tools/sketch_bench.py > "$RAW/sketch_bench.log" 2>&1 && cp gpurun_out/sketch_bench.json "$OUT/${R}_sketch_bench.json"
for v in "fp32 gaussian" "fp32 gaussian --torch-sketch" "fp32 gaussian --torch-sketch --sketch-bf16" "fp32 rademacher" "bf16 gaussian" "bf16 gaussian --torch-sketch" "bf16 rademacher" "fp32 dct" "bf16 dct" "fp32 dft" "bf16 dft"; do
    set -- $v; dt=$1; mm=$2; shift 2; tag=$(echo "$dt $mm $@" | tr " " "_" | tr -d "-" | sed 's/_$//')
    timeout 600 python3 tools/roberta_bench.py --table --dtype $dt --matmul $mm --steps 6 "$@" 2>> "$RAW/roberta.err" | tail -1 > "$OUT/${R}_roberta_table_$tag.json"
done
# the ratio the reference quotes (0.2: p = 3276 of 16384 rows), both layer widths of RoBERTa-base, both sketches: >= 200 settled
# dispatches each (writes gpurun_out/profiles_$R/${R}_sketch_rocprof_*_p3276_bf16.txt itself)
bash tools/profile_sketch.sh $R > "$RAW/profile_sketch.log" 2>&1
# the sampled cosine transform (fewbit_hip_sampled_dct) at the same shapes: settled kernel durations and PMC traffic
bash tools/profile_dct.sh $R > "$RAW/profile_dct.log" 2>&1
# the randomized RoBERTa step with the arms interleaved in ONE process (S from memory / fused / fp32 partial sums / Rademacher),
# where its GPU time goes by kernel class, and the counters behind the in-situ forward (EXPERIMENTS.md 5)
for dt in fp32 bf16; do timeout 900 python3 tools/roberta_ab.py $dt 3 2>&1 | grep -v amdgpu.ids > "$OUT/${R}_roberta_ab_$dt.txt"; done
bash tools/profile_insitu_sketch.sh $R > "$RAW/insitu_sketch.log" 2>&1 && cp gpurun_out/${R}_roberta_randomized_insitu.json "$OUT/"
[ -x scratch/bin/gen_bench ] && scratch/bin/gen_bench > "$OUT/${R}_gen_bench.txt" 2>&1
python3 scratch/timeline.py 0.0 2>&1 | grep -v amdgpu.ids > "$OUT/${R}_clock_transient_timeline.txt"
[ -f gpurun_out/fp32_ulp.json ] && cp gpurun_out/fp32_ulp.json "$OUT/${R}_fp32_ulp.json"   # written by pytest -m gpu
ls -la "$OUT"
