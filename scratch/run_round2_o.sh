#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -q -k "concurrent" 2>&1 | tail -2
FEWBIT_HIP_LUT_MIN=1 python scratch/xover.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02o_xover_table.log
FEWBIT_HIP_LUT_MIN=999999999999 python scratch/xover.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02o_xover_search.log
paste gpurun_out/r02o_xover_table.log gpurun_out/r02o_xover_search.log
