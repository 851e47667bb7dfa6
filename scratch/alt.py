import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from fewbit_amd import cabi
from tests.helpers import from_raw
z = np.load('tests/golden/quantize_ref.npz'); dev = 'cuda'; n = 4096 * 4096
b = from_raw(z['gelu03_bf16_borders'], torch.bfloat16).to(dev); l = from_raw(z['gelu03_bf16_levels'], torch.bfloat16).to(dev)
def mk():
    x = torch.randn(n, device=dev).to(torch.bfloat16); y = torch.empty_like(x)
    st = torch.empty(cabi.state_nbytes(n, 3), dtype=torch.uint8, device=dev)
    return x, y, st
sets = [mk() for _ in range(8)]
gy = torch.randn(n, device=dev).to(torch.bfloat16); gx = torch.empty_like(gy)
def timeit(fns, iters=400):
    for _ in range(3):
        for f in fns: f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        for f in fns: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters
F = [cabi.bind_forward('gelu', x, b, out=y, state=st) for x, y, st in sets]
B = [cabi.bind_backward(gy, st, l, out=gx) for x, y, st in sets]
x0, y0, st0 = sets[0]
Bfull = [cabi.bind_backward(y, st, l, out=x) for x, y, st in sets]   # distinct gy/gx per set (reuse y as gy, x as gx)
print('fwd same buffers            : %.2f us/launch' % timeit([F[0]]))
print('fwd rotating 2 sets (140MiB): %.2f us/launch' % (timeit([F[0], F[1]]) / 2))
print('fwd rotating 4 sets (280MiB): %.2f us/launch' % (timeit([F[0], F[1], F[2], F[3]]) / 4))
print('fwd rotating 8 sets (560MiB): %.2f us/launch' % (timeit(F) / 8))
print('bwd same buffers            : %.2f us/launch' % timeit([B[0]]))
print('bwd rotating 8 sets         : %.2f us/launch' % (timeit(Bfull) / 8))
print('fwd+bwd same set            : %.2f us/step' % timeit([F[0], B[0]]))
print('fwd+bwd rotating 8 sets     : %.2f us/step' % (timeit([f for pair in zip(F, Bfull) for f in pair]) / 8))
copyk = lambda: y0.copy_(x0)
print('fwd + torch copy            : %.2f us/pair (copy alone %.2f)' % (timeit([F[0], copyk]), timeit([copyk])))
