#!/bin/bash
# round 6, call q: rows = 2^17 and 2^18 on the sampled-DCT kernel pair (512-point tiles): tests, times against torch.fft
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_dct.py tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -8 | cut -c1-400
OUT=gpurun_out/r06_dct_rows_big.txt
echo "# sampled DCT at rows = 2^17, 2^18 (512-point tiles of 128 KiB, one workgroup per CU): tools/dct_run.py <rows> 768 <rows/5> bf16 50 30 seeded|torch  (HIP events, settled 30 ms)" > $OUT
for rows in 131072 262144 65536; do
  for mode in seeded torch; do
    reps=50; [ $mode = torch ] && reps=10
    timeout 200 python3 tools/dct_run.py $rows 768 $((rows / 5)) bf16 $reps 30 $mode 2>/dev/null | tail -1 >> $OUT
  done
done
timeout 200 python3 tools/dct_run.py 131072 3072 26214 bf16 20 30 seeded 2>/dev/null | tail -1 >> $OUT
timeout 200 python3 tools/dct_run.py 131072 768 26214 f32 50 30 seeded 2>/dev/null | tail -1 >> $OUT
cut -c1-260 $OUT
