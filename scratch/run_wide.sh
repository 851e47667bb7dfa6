timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python scratch/generic_perf.py
timeout 200 python bench.py --steps 1000 --no-cpu-baseline | cut -c1-120
