"""soak failure: gaussian bf16 1 x 8 (ld 11) proj 33 on the 128 x 512 tile"""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import sketch_reference as ref
from fewbit_amd import cabi
DEV = 'cuda:0'
for rows, features, proj, ld in ((1, 8, 33, 11), (1, 8, 33, 8), (1, 8, 33, 16), (2, 8, 33, 11), (1, 16, 33, 19), (7, 8, 33, 11), (1, 8, 1, 11), (1, 8, 300, 11)):
    g = torch.Generator().manual_seed(rows * 31 + features)
    m = torch.randn(rows, ld, generator=g).to(torch.bfloat16)[:, :features]
    md = m.to(DEV)
    for seed in (5, 0xabcdef0123456789):
        S = ref.matrix('gaussian', seed, proj, rows, torch.bfloat16).double()
        want = S @ m.double()
        for halves, waves in itertools.product((1, 2), (-1, 4, 8)):
            cabi.tune_sketch_halves(halves); cabi.tune_sketch_waves(waves); cabi.tune_sketch_materialise(0)
            got = cabi.sketch('gaussian', md, proj, seed, 1.0).cpu().double()
            err = (got - want).abs()
            bad = (err > 0.02 * want.abs() + 1e-2) | ~torch.isfinite(got)
            print(rows, features, proj, 'ld', ld, 'stride', md.stride(), 'ptr%16', md.data_ptr() % 16, 'halves', halves, 'waves', waves, 'seed', hex(seed)[:6], 'bad', int(bad.sum()), 'nan', int((~torch.isfinite(got)).sum()),
                  'maxerr', float(err[torch.isfinite(err)].max()) if torch.isfinite(err).any() else None, flush=True)
            if bad.any():
                idx = bad.nonzero()[:4].tolist()
                print('    first bad', idx, [(float(got[i, j]), float(want[i, j])) for i, j in idx])
