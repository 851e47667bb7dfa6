#!/bin/bash
# round 5, call ao: the fp32 rows of the RoBERTa table and the in-situ kernel-class breakdown under the width rule
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
bash scratch/box_fingerprint.sh | grep -i "vbios_version\|smc\|MEC firm" | head -4 > gpurun_out/r05ao_box.txt
for v in "fp32 gaussian" "fp32 rademacher"; do set -- $v
    timeout 600 python3 tools/roberta_bench.py --table --dtype $1 --matmul $2 --steps 6 2> gpurun_out/r05ao_roberta.err | tail -1 > gpurun_out/r05ao_roberta_table_$1_$2.json
done
for dt in fp32; do timeout 900 python3 scratch/roberta_ab.py $dt 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r05ao_roberta_ab_$dt.txt; done
bash tools/profile_insitu_sketch.sh r05 > gpurun_out/r05ao_insitu.log 2>&1
cat gpurun_out/r05ao_box.txt gpurun_out/r05ao_roberta_ab_fp32.txt; cut -c1-400 gpurun_out/r05ao_roberta_table_fp32_gaussian.json
