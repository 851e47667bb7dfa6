#!/usr/bin/env python3
"""Re-export the reference's built-in quantization tables into fewbit_amd/data/builtin.npz.

The tables are DATA (13 functions x 1..4 bits x {borders, levels}, float64), produced offline by the
reference's `fewbit quantize` recipe (tools/quantize-builtins.sh there) and shipped by it as
fewbit/data/builtin.npz.  Parity of the packed codes and gradients is only meaningful on the very same
constants, so they are copied value-for-value (same key scheme `{func}{bits:02d}-borders|levels`, which is
also the on-disk format StepwiseStore.load reads, fewbit/functional/activations.py:69-81).
Run in the build container only:  python tools/export_tables.py
"""
from pathlib import Path

import numpy as np

SRC = Path('/root/reference/fewbit/data/builtin.npz')
DST = Path(__file__).resolve().parents[1] / 'fewbit_amd' / 'data' / 'builtin.npz'

with np.load(SRC) as z:
    tables = {k: np.asarray(z[k], dtype=np.float64) for k in sorted(z.files)}
for k, v in tables.items():
    assert v.ndim == 1 and np.isfinite(v).all(), k
DST.parent.mkdir(parents=True, exist_ok=True)
np.savez_compressed(DST, **tables)
print(f'{len(tables)} arrays -> {DST} ({DST.stat().st_size} bytes)')
