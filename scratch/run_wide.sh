timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 300 python scratch/generic_perf.py
