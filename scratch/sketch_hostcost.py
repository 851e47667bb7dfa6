"""host time per call of the sketch path (tiny tensors: the GPU is never the bottleneck)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fewbit
from fewbit_amd import cabi, linear

def per_call(f, n=2000):
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

m = torch.randn(256, 64, device='cuda', dtype=torch.bfloat16)
out = torch.empty(32, 64, device='cuda', dtype=torch.bfloat16)
print(f'cabi.sketch (allocating out)            {per_call(lambda: cabi.sketch("rademacher", m, 32, 1)):6.1f} us')
print(f'cabi.sketch (out given)                 {per_call(lambda: cabi.sketch("rademacher", m, 32, 1, out=out)):6.1f} us')
print(f'torch.matmul of the same shape          {per_call(lambda: torch.matmul(out[:, :32].float(), out[:32].float())):6.1f} us')
print(f'linear._draw_seed(None)                 {per_call(lambda: linear._draw_seed(None)):6.1f} us')
lin = fewbit.RandomizedLinear(64, 64, proj_dim=32, matmul='rademacher', device='cuda', dtype=torch.bfloat16)
ref = torch.nn.Linear(64, 64, device='cuda', dtype=torch.bfloat16)
x = torch.randn(256, 64, device='cuda', dtype=torch.bfloat16, requires_grad=True)
def step(layer):
    y = layer(x)
    y.sum().backward()
print(f'RandomizedLinear fwd+bwd (native sketch) {per_call(lambda: step(lin), 500):6.1f} us')
print(f'nn.Linear fwd+bwd                        {per_call(lambda: step(ref), 500):6.1f} us')
linear.use_native_sketch(False)
print(f'RandomizedLinear fwd+bwd (torch sketch)  {per_call(lambda: step(lin), 500):6.1f} us')
