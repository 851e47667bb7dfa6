#!/bin/bash
# round 6, call f: the final sampled-DCT kernels: their tests, the settled profile with PMC traffic, the sketch bench with the dct column, the RoBERTa rows
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_dct.py tests/test_gpu_linear.py tests/test_api.py -q 2>&1 | tail -3
bash tools/profile_dct.sh r06 > gpurun_out/r06f_profile_dct.log 2>&1; grep -E "sum of|pass_[ab]:" gpurun_out/r06f_profile_dct.log
timeout 1200 python3 tools/sketch_bench.py > gpurun_out/r06f_sketch_bench.log 2>&1; cp gpurun_out/sketch_bench.json gpurun_out/profiles_r06/r06_sketch_bench.json
for dt in bf16 fp32; do
    timeout 600 python3 tools/roberta_bench.py --table --dtype $dt --matmul dct --steps 6 2>/dev/null | tail -1 > gpurun_out/profiles_r06/r06_roberta_table_${dt}_dct.json
    timeout 900 python3 tools/roberta_ab.py $dt 3 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_r06/r06_roberta_ab_$dt.txt; tail -9 gpurun_out/profiles_r06/r06_roberta_ab_$dt.txt | cut -c1-200
done
