#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r02k_pytest.log 2>&1
tail -6 gpurun_out/r02k_pytest.log
SIZES=16777216,33554432,50331648,67108864,134217728 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02k_headvar.log
DT=f32 SIZES=16777216,50331648 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02k_headvar.log
python scratch/headvar.py 4 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02k_headvar.log
python tools/roberta_bench.py --dtype fp32 > gpurun_out/r02k_roberta_fp32.json 2> gpurun_out/r02k_roberta.err; cat gpurun_out/r02k_roberta_fp32.json | cut -c1-900
python tools/roberta_bench.py --dtype bf16 > gpurun_out/r02k_roberta_bf16.json 2>> gpurun_out/r02k_roberta.err; cat gpurun_out/r02k_roberta_bf16.json | cut -c1-900
