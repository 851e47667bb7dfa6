#!/bin/bash
# round 6, call m: rows = 3 x 2^k on the sampled-DCT kernel pair: tests, the 1500-case soak, time against torch.fft
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_dct.py -q -m gpu -x 2>&1 | tail -12 | cut -c1-400
OUT=gpurun_out/r06_dct_rows_3x.txt
echo "# sampled DCT at rows = 3 x 2^k (radix-3 first stage in pass B): tools/dct_run.py <rows> 768 <rows/5> bf16 100 30 seeded|torch  (HIP events, settled 30 ms)" > $OUT
for rows in 768 3072 6144 12288 24576 49152 8192 16384; do
  for mode in seeded torch; do
    reps=100; [ $mode = torch ] && reps=20
    timeout 120 python3 tools/dct_run.py $rows 768 $((rows / 5)) bf16 $reps 30 $mode 2>/dev/null | tail -1 >> $OUT
  done
done
timeout 120 python3 tools/dct_run.py 12288 3072 2457 bf16 100 30 seeded 2>/dev/null | tail -1 >> $OUT
timeout 120 python3 tools/dct_run.py 12288 768 2457 f32 100 30 seeded 2>/dev/null | tail -1 >> $OUT
cut -c1-260 $OUT
