#!/usr/bin/env python3
"""profiles/README.md = this table: file pattern -> one line -> where it is quoted.  `python tools/profiles_index.py` rewrites the README and
fails if a file under profiles/ matches no row (tests/test_api.py runs the check).  rNN = any round that has the file; the newest is the one quoted."""
import fnmatch
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
D, E = 'DESIGN', 'EXPERIMENTS'
ROWS = [
    # ---- the round's core set (tools/profile_round.sh)
    ('r0?_bench_line.json', 'python bench.py: the driver-visible line with every extra', f'{D} 7.1, 7.2'),
    ('r0?_bench_line_k20.json', 'bench.py --steps 20 --warmup 5 --no-extras: the driver\'s short shape', f'{D} 7.1'),
    ('r0?_bench_line_c4.json', 'bench.py --config c4: the 8192 x 4096 shard of BASELINE config 4', f'{D} 7.2'),
    ('r0?_bench_line_final.json|r0?_bench_line_live_pmc.json', 'further runs of the default line (round 4 / 5: final tree, first run with live PMC traffic)', f'{E} 5'),
    ('r0?_bench_line_2ranks_*.json|r0?_bench_line_?ranks_c?_shared.json', 'bench.py --gpus N with the ranks sharing one GPU: the launch paths of an 8-GPU lease', f'{D} 8'),
    ('r0?_strong_*ranks_one_gpu.json', '--scaling strong: one tensor cut over N ranks sharing the GPU', f'{D} 8'),
    ('r0?_strong_*_emulated_world_?.json', "--emulate-world M: rank 0's slice of an M-way cut alone on the GPU + the projected M-GPU figure", f'{D} 8'),
    ('r0?_bench_kernel_stats.csv', 'rocprofv3 --kernel-trace --stats over bench.py --no-extras: per-dispatch averages of the two kernels', f'{D} 7.1'),
    ('r0?_pmc_traffic.json|traffic_forward.json|r01_pmc_*_counter_collection.csv', 'PMC FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes: HBM bytes per launch', f'{D} 7.1'),
    ('r0?_configs.json', 'every BASELINE config warm and cold: rocprofv3 per-dispatch durations + PMC traffic per phase', f'{D} 7.2'),
    ('r0?_roberta_base*.json|r01_roberta_readme_table.json', 'RoBERTa-base b128 x s128: vanilla, fewbit.GELU route, raw-operator route', f'{D} 7.2'),
    ('r0?_roberta_insitu.json|r0?_roberta_kernel_stats_*.csv', 'rocprofv3 kernel stats of the RoBERTa step: the few-bit kernels inside training', f'{D} 7.2'),
    ('r0?_hostcost.json|r0?_oplevel_graph.txt|r0?_wall_overhead.txt', 'host cost per operator call; hipGraph replay; wall-clock overhead of a 20-step region', f'{E} 5'),
    ('r0?_clock_transient_timeline.txt', 'step time against time since idle: the transient every settled figure waits out', f'{D} 7'),
    ('r0?_stream_bench_*MiB.txt', 'bare copy / read / write kernels and an empty kernel: the floors', f'{D} 7.1, 9'),
    ('r0?_fp32_ulp.json', 'ULP histogram of the fp32 GELU forward against the reference run and the exact value', f'{D} 6'),
    ('r0?_soak_fuzz.txt|r0?_sketch_soak_fuzz.txt|r06_dct_soak_fuzz.txt', 'one-off soak runs of the differential fuzz tests', f'{D} 6'),
    # ---- estimators of the randomized layers
    ('r0?_sketch_bench*.json', 'tools/sketch_bench.py: every estimator per shape against torch (round 6: settled, with the dct / dft columns; _torchfft_dct: before the DCT kernel existed)', f'{D} 7.3'),
    ('r0?_sketch_rocprof_*.txt', 'tools/profile_sketch.sh: rocprofv3 kernel durations (round 6: >= 200 settled dispatches) + PMC counters of the dense sketches', f'{D} 7.3'),
    ('r06_dct_rocprof_*.txt', 'tools/profile_dct.sh: settled kernel durations and PMC traffic of the sampled-DCT kernel pair', f'{D} 5, 7.3'),
    ('r06_dct_large_rows.txt', 'the kernel pair and the torch.fft formulation at 32768 and 65536 rows', f'{D} 7.3'),
    ('r06_convergence_demo_*.txt', 'a student MLP fitted with torch layers and with each few-bit configuration (tools/convergence_demo.py); the DCT estimator at lr 1.0', f'{D} 7.4'),
    ('r06_dct_rows_big.txt', 'the kernel pair and the torch.fft formulation at 2^17 and 2^18 rows (512-point tiles)', f'{D} 7.3'),
    ('r06_dct_rows_5x.txt', 'the kernel pair and the torch.fft formulation at 5 x 2^k rows (1280 .. 40960)', f'{D} 7.3'),
    ('r06_dct_rows_3x.txt', 'the kernel pair and the torch.fft formulation at 3 x 2^k rows (768 .. 49152)', f'{D} 7.3'),
    ('r06_dct_variants.txt', 'the sampled-DCT variants measured in round 6, phases compiled out, per-workgroup timeline', f'{D} 7.5'),
    ('r06_dct_sorted_samples.txt', 'the samples sorted by residue class once, in pass A, instead of tested by every pass-B workgroup: pass B 17.3 -> 12.7 us', f'{D} 5'),
    ('r06_dct_serve_lanes.txt', 'pass B writing a sampled row with 16 / 8 / 4 lanes: 4 shipped (pass B -8 %)', f'{D} 7.5'),
    ('r06_dct_rounds.txt', 'both DCT passes against the number of workgroups (features swept): a fixed 7-9 us plus 7.5 ns per workgroup', f'{D} 7.5'),
    ('r06_dct_stagger.txt|r06_dct_fused_upper_bound.txt|r06_dct_inter16.txt', 'DCT experiments not kept: staggered starts, both passes in one launch (timing only), a bf16 intermediate', 'EXPERIMENTS.md'),
    ('r0?_roberta_table_*.json', "tools/roberta_bench.py --table: the reference README's RoBERTa table per dtype and estimator", f'{D} 7.4'),
    ('r0?_roberta_ab_fp32.txt|r0?_roberta_ab_bf16.txt|r0?_roberta_randomized_insitu*.json', 'the randomized RoBERTa step, arms interleaved in one process; its GPU time by kernel class', f'{D} 7.4'),
    ('r06_roberta_overlap_ab.txt', 'the estimators on a side stream beside the layer GEMMs: slower, not kept', f'{E} round 6'),
    ('r06_isa_identity.txt', 'tools/isa_digest.py: machine code of all 628 device functions before / after the round-6 source clean-up', f'{D} 9'),
    # ---- experiments (kept as records; the design quotes only their conclusions)
    ('r02_launch_shape_sweep_*.txt|r03_*shape_sweep*.txt|r03_backward_size_crossover.txt', 'launch shape, groups per lane and size crossovers of the activation kernels', f'{D} 3.1'),
    ('r03_ablation_*.txt|r03_kernels_r02_vs_r03_ab.txt|r03_inplace_stores.txt|r04_inplace_*.txt', 'one stage compiled out at a time; in-place stores; round-to-round A/B', f'{D} 7.1; {E} 6'),
    ('r02_forward_head_variants.txt|r02_fp32_split_layout_ab.txt|r02_wave_timeline_trace.txt|r03_wave_priority_and_early_head.txt|r03_saddr_addressing_ab.txt|r04_state_staging_ab.txt',
     'forward head variants, fp32 split layout, wave timelines, s_setprio, saddr addressing, staged state stores', f'{E} 6'),
    ('r04_insitu_*.txt|r05_insitu_pmc*.txt', 'the forward right behind the GEMM that feeds it: time, launch shape, counters', f'{D} 7.2; {E} 5'),
    ('r0?_all_fp32_*.txt|r0?_all_patterns_more.txt|r0?_ragged_sweep.txt', 'exhaustive sweeps: all 2^32 fp32 patterns, every (gy, level) product, every remainder', f'{D} 6'),
    ('r04_injected_child_failure.txt', "a failing rank's traceback reaches the parent", f'{D} 8'),
    ('r04_sketch_balance.txt|r04_sketch_hostcost.txt|r04_sketch_preconvert.txt', 'slice balance, host cost and the conversion pass of the dense sketches', f'{D} 4'),
    ('r05_gen_bench*.txt|r0?_gaussian_ablation.txt|r05_sketch_*_ab.txt|r06_gen_bench.txt', 'generator cost beside an MFMA stream; A/B of the sketch data paths', f'{D} 4; {E} 3.1'),
    ('r05_roberta_ab_*.txt|r05_roberta_power_*.txt|r05_box_fingerprints.txt|r05_roberta_table_*_width_rule.json',
     'round 5: the width rule of fp32 input and the GPU-pool forensics behind it (closed; not to be repeated)', f'{D} 4, 9; {E} 5.1'),
    ('README.md', 'this index', ''),
]


def covered(name: str) -> bool:
    return any(fnmatch.fnmatch(name, pat) for row in ROWS for pat in row[0].split('|'))


def main() -> int:
    files = sorted(p.name for p in (ROOT / 'profiles').iterdir() if p.is_file())
    missing = [f for f in files if not covered(f)]
    if missing:
        sys.stderr.write('files under profiles/ without a row in tools/profiles_index.py: ' + ', '.join(missing) + '\n')
        return 1
    if len(sys.argv) > 1 and sys.argv[1] == '--check':
        return 0
    lines = ['Evidence kept per round (`rNN_*`; a round\'s core set is regenerated by `bash tools/profile_round.sh rNN` through `gpurun`). Generated by',
             '`python tools/profiles_index.py`, which fails when a file has no row. Where several rounds hold the same file the newest is the one quoted.', '',
             '| file(s) | what it is | quoted in |', '|---|---|---|']
    for pats, what, where in ROWS:
        n = sum(1 for f in files if any(fnmatch.fnmatch(f, p) for p in pats.split('|')))
        lines.append(f"| {', '.join('`' + p + '`' for p in pats.split('|'))} ({n}) | {what} | {where} |")
    (ROOT / 'profiles' / 'README.md').write_text('\n'.join(lines) + '\n')
    return 0


if __name__ == '__main__':
    sys.exit(main())
