#!/bin/bash
# round 6, call c: the native sampled-DCT kernel pair: its parity tests, the whole GPU suite, its settled rocprofv3 profile with PMC traffic, the sketch
# bench (dct column now the kernel pair) and the RoBERTa-base table with --matmul dct
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_dct.py -m gpu -q 2>&1 | tail -40 | cut -c1-400 > gpurun_out/r06c_pytest_dct.txt; tail -25 gpurun_out/r06c_pytest_dct.txt
timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -25 | cut -c1-400 > gpurun_out/r06c_pytest_gpu.txt; tail -8 gpurun_out/r06c_pytest_gpu.txt
bash tools/profile_dct.sh r06 > gpurun_out/r06c_profile_dct.log 2>&1; tail -60 gpurun_out/r06c_profile_dct.log | cut -c1-300
timeout 1200 python3 tools/sketch_bench.py > gpurun_out/r06c_sketch_bench.log 2>&1; cp gpurun_out/sketch_bench.json gpurun_out/r06c_sketch_bench.json
for dt in bf16 fp32; do
    timeout 600 python3 tools/roberta_bench.py --table --dtype $dt --matmul dct 2>gpurun_out/r06c_roberta_${dt}_dct.err | tail -1 > gpurun_out/r06c_roberta_table_${dt}_dct.json
    python3 -c "
import json
d=json.load(open('gpurun_out/r06c_roberta_table_${dt}_dct.json'))
print('$dt', d['config'][-120:], [(r['gelu'],r['linear'],r['ms_per_step'],r['step_time_ratio'],r['saving_pct']) for r in d['rows']])" 2>&1 | tail -1
done
python3 - <<'PY'
import json
for r in json.load(open('gpurun_out/r06c_sketch_bench.json')):
    st = r.get('sampled_transform')
    if st: print(r['dtype'], r['features'], r['proj'], 'rademacher', [v['us'] for k, v in r.items() if k.startswith('rademacher_')], 'dct', st['dct']['us'], st['dct']['x_byte_floor'], 'dft', st['dft']['us'])
PY
