import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT); sys.path.insert(0, ROOT + '/tests')
import numpy as np, torch
from fewbit_amd import cabi
from fewbit_amd.store import store
from helpers import ulp_distance
from test_gpu_numerics import _exact, all_16bit
for name, p in (('elu', (0.7,)), ('celu', (1.3,)), ('selu', ()), ('tanhshrink', ())):
    for dtype in (torch.bfloat16, torch.float16):
        x = all_16bit(dtype); b, _ = store.get(name, 3, 'cuda', dtype)
        y, _ = cabi.quantize_forward(name, x.cuda(), b[1:-1].contiguous(), *p); y = y.cpu()
        ex64 = torch.from_numpy(_exact(name, x.double().numpy(), p)); ex = ex64.to(dtype)
        fin = torch.isfinite(x) & torch.isfinite(ex64)
        d = ulp_distance(y, ex); tiny = (y.double().abs() <= 1e-36) & (ex64.abs() <= 1e-36)
        bad = fin & ~((d <= 1) | tiny)
        print(name, dtype, 'bad', int(bad.sum()), 'exact-match', float((d[fin] == 0).double().mean()))
        if bad.any():
            i = bad.nonzero().flatten()
            sel = i[:: max(1, len(i) // 8)][:8]
            print('  x', x[sel].float().tolist()); print('  y', y[sel].float().tolist()); print('  e', ex[sel].float().tolist()); print('  d', d[sel].tolist())
