import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch, fewbit
from fewbit_amd import cabi
DEV='cuda'
for dtype in (torch.float16, torch.bfloat16, torch.float32):
    x = torch.zeros(5009, dtype=dtype); x[5001] = -0.0; x[100] = -0.0
    b = torch.tensor([0.5, 1.0, 2.0]).to(dtype)
    for fn in ('identity', 'identity_fold'):
        y, st = cabi.quantize_forward(fn, x.to(DEV), b.to(DEV), 0.0)
        print(dtype, fn, 'cabi', y[100].item(), y[5001].item(), torch.signbit(y[[100, 5001]]).tolist())
    xd = x.to(DEV)
    z = xd * 1.0
    print('mul', torch.signbit(z[[100, 5001]]).tolist())
    l = torch.tensor([1.0, 0.6, 0.3, 0.1]).to(dtype).to(DEV)
    out = torch.ops.fewbit.stepwise_folded(z, b.to(DEV), l, True, 0.0, 0.0)
    print('op', torch.signbit(out[[100, 5001]]).tolist())
