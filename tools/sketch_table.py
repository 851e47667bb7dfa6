#!/usr/bin/env python3
"""markdown rows of the estimator table (the full sweep: EXPERIMENTS.md 5.1) from tools/sketch_bench.py's JSON:  python tools/sketch_table.py profiles/r06_sketch_bench.json"""
import json
import sys

rows = json.load(open(sys.argv[1]))
print('| input | M (rows × features) | p | Rademacher: time = TFLOP/s | Gaussian | torch: randint+mm / randn+mm | torch.matmul alone | grid (x, y, slices): Rademacher / Gaussian |')
print('|---|---|---|---|---|---|---|---|')
for r in rows:
    rk = next(k for k in r if k.startswith('rademacher_'))
    gk = next(k for k in r if k.startswith('gaussian_'))
    ra, ga = r[rk], r[gk]
    dt = {'bfloat16': 'bf16', 'float32': 'fp32', 'float16': 'fp16'}[r['dtype']]
    win = '' if ga['us'] <= r['torch_randn_plus_matmul_us'] else ' (slower than torch)'
    print(f"| {dt} | {r['rows']} × {r['features']} | {r['proj']} | {ra['us']:.0f} µs = **{ra['TFLOPs']:.0f}** ({ra['frac_of_bf16_peak']:.2f}) | "
          f"{ga['us']:.0f} µs = {ga['TFLOPs']:.0f} ({ga['frac_of_bf16_peak']:.2f}){win} | {r['torch_randint_plus_matmul_us']:.0f} / {r['torch_randn_plus_matmul_us']:.0f} µs | "
          f"{r['torch_matmul_only_us']:.0f} µs = {r['torch_matmul_only_TFLOPs']:.0f} | {ra['grid']} / {ga['grid']} |")
