#!/bin/bash
# round 6, call g: the native estimators on a side stream beside the layer's GEMMs (FEWBIT_SKETCH_OVERLAP): tests, then the RoBERTa arms with it off and on
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_linear.py tests/test_gpu_dct.py tests/test_gpu_sketch.py tests/test_gpu_roberta.py -q -x 2>&1 | tail -4 | cut -c1-300
for dt in bf16 fp32; do
  for ov in 0 1; do
    echo "== $dt FEWBIT_SKETCH_OVERLAP=$ov"
    FEWBIT_SKETCH_OVERLAP=$ov timeout 900 python3 tools/roberta_ab.py $dt 3 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-200
  done
done
