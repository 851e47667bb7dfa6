#!/bin/bash
# round 6, call n: the whole GPU suite and the DCT soak on the tree with 3 x 2^k rows
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -4 | cut -c1-300
OUT=gpurun_out/r06_dct_soak_fuzz.txt
echo "FEWBIT_DCT_FUZZ_CASES=1500 python -m pytest tests/test_gpu_dct.py -q -k fuzz   (MI355X, round 6 final kernels: rows 2^8..2^16, 3 x 2^8..3 x 2^14 and 5 x 2^8..5 x 2^13 drawn at random, 2^17 and 2^18 among the fixed cases, ragged / odd feature counts, p from 1 to rows, three dtypes, strides, scales, against the float64 DCT-II on the device; per case fewbit_hip_sampled_dct_seeded bit-equal to the explicit call on fewbit_hip_sampled_rows of the same seed; + the list-overflow and p > 4096 regimes)" > $OUT
FEWBIT_DCT_FUZZ_CASES=1500 timeout 900 python3 -m pytest tests/test_gpu_dct.py -q -k fuzz 2>&1 | tail -3 >> $OUT; cut -c1-200 $OUT
