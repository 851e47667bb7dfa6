"""which settings break gaussian fp16 65 x 511 proj 31 (fuzz failure of run c)?"""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import sketch_reference as ref
from fewbit_amd import cabi
DEV = 'cuda:0'
for dist, dtype, rows, features, proj in (('gaussian', torch.float16, 65, 511, 31), ('gaussian', torch.bfloat16, 65, 511, 31), ('gaussian', torch.float16, 65, 512, 31), ('rademacher', torch.float16, 65, 511, 31), ('gaussian', torch.float16, 300, 263, 40)):
    g = torch.Generator().manual_seed(rows * 31 + features)
    for ldx in (0, 3, 8):
        m = torch.randn(rows, features + ldx, generator=g).to(dtype)[:, :features]
        S = ref.matrix(dist, 5, proj, rows, dtype).double()
        want = S @ m.double()
        for waves, halves, slices, mem in itertools.product((-1, 4, 8), (1, 2), (-1, 2), (0, 1)):
            cabi.tune_sketch_waves(waves); cabi.tune_sketch_halves(halves); cabi.tune_sketch_slices(slices); cabi.tune_sketch_materialise(mem)
            got = cabi.sketch(dist, m.to(DEV), proj, 5, 1.0).cpu().double()
            err = (got - want).abs()
            bad = (err > 0.05 * want.abs() + 0.5).nonzero()
            if len(bad):
                cols = sorted(set(bad[:, 1].tolist())); rws = sorted(set(bad[:, 0].tolist()))
                print(dist, dtype, rows, features, proj, 'ld+', ldx, 'waves', waves, 'halves', halves, 'slices', slices, 'mem', mem, cabi.describe_sketch(dist, rows, features, proj, dtype)['kernel'],
                      'BAD cols', cols[:6], '..', cols[-3:], 'rows', rws[:4], '..', rws[-2:], 'n', len(bad), flush=True)
print('done')
