#!/bin/bash
# round 6, call l: the DCT soak with the seeded entry point checked per case, the C-ABI demo, smoke()
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r06_dct_soak_fuzz.txt
echo "FEWBIT_DCT_FUZZ_CASES=1500 python -m pytest tests/test_gpu_dct.py -q -k fuzz   (MI355X, round 6 final kernels: rows 2^8..2^16, ragged / odd feature counts, p from 1 to rows, three dtypes, strides, scales, against the float64 DCT-II on the device; per case fewbit_hip_sampled_dct_seeded bit-equal to the explicit call on fewbit_hip_sampled_rows of the same seed; + the list-overflow and p > 4096 regimes)" > $OUT
FEWBIT_DCT_FUZZ_CASES=1500 timeout 900 python3 -m pytest tests/test_gpu_dct.py -q -k fuzz 2>&1 | tail -3 >> $OUT; cat $OUT | cut -c1-200
timeout 600 python3 -m pytest tests/test_gpu_dct.py tests/test_gpu_ops.py -q -m gpu 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
./examples/cabi_demo | tail -4
