#!/bin/bash
# round 6, call r: rows = 5 x 2^k on the sampled-DCT kernel pair (radix-5 first stage in pass B): tests, times against torch.fft
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_dct.py tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -6 | cut -c1-400
FEWBIT_DCT_FUZZ_CASES=1500 timeout 900 python3 -m pytest tests/test_gpu_dct.py -q -k fuzz 2>&1 | tail -2
OUT=gpurun_out/r06_dct_rows_5x.txt
echo "# sampled DCT at rows = 5 x 2^k (radix-5 first stage in pass B): tools/dct_run.py <rows> 768 <rows/5> bf16 100 30 seeded|torch  (HIP events, settled 30 ms)" > $OUT
for rows in 1280 5120 10240 20480 40960; do
  for mode in seeded torch; do
    reps=100; [ $mode = torch ] && reps=20
    timeout 120 python3 tools/dct_run.py $rows 768 $((rows / 5)) bf16 $reps 30 $mode 2>/dev/null | tail -1 >> $OUT
  done
done
cut -c1-230 $OUT
