#!/bin/bash
# first GPU pass of round 2: full GPU test suite, the new bench line (default and at the driver's short step count), and
# the forward-head variants
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02a_pytest.log 2>&1
tail -5 gpurun_out/r02a_pytest.log
python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; tail -c 600 gpurun_out/r02a_bench.err
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r02a_bench_k20.json 2>> gpurun_out/r02a_bench.err
python bench.py --steps 200 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r02a_bench_k200.json 2>> gpurun_out/r02a_bench.err
python bench.py --config c4 --no-extras --no-cpu-baseline > gpurun_out/r02a_bench_c4.json 2>> gpurun_out/r02a_bench.err
for v in default h1 h2 h3 h5 h7; do
  if [ $v = default ]; then unset FEWBIT_HIP_LIB; else export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_$v.so; fi
  python scratch/headvar.py 3 >> gpurun_out/r02a_headvar.log 2>&1
done
unset FEWBIT_HIP_LIB
cat gpurun_out/r02a_headvar.log
for f in gpurun_out/r02a_bench_k20.json gpurun_out/r02a_bench_k200.json gpurun_out/r02a_bench_c4.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['timing']['wall_ms_per_step'], d['roofline']['frac'])"; done
python -c "import json; d=json.load(open('gpurun_out/r02a_bench.json')); print(json.dumps({k:d[k] for k in ('value','ms_per_step','pct_of_hbm_roofline','fwd_us','bwd_us','roofline','cold','cpu_baseline','cpu_baseline_1thread')},indent=0)); print(json.dumps(d['configs'],indent=0))"
