"""Host-side cost per call of the op-level path (module -> functional -> torch.ops -> autograd -> C-ABI) for tensors
small enough that the GPU is never the bottleneck, and the op-level fwd+bwd time at the headline size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch, fewbit
import fewbit.functional as F
dev = 'cuda'
def host(f, iters=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): f()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / iters * 1e6
x = torch.randn(1024, device=dev, dtype=torch.bfloat16)
xr = x.clone().requires_grad_(True)
act = fewbit.GELU(bits=3); relu = fewbit.ReLU(); van = torch.nn.GELU()
b, l = F.store.get('gelu', 3, dev, torch.bfloat16)
bi = b[1:-1].contiguous()
g = torch.ones_like(x)
print('torch.nn.GELU fwd (no grad)        %.1f us' % host(lambda: van(x)))
print('torch.ops.fewbit.gelu (no grad)    %.1f us' % host(lambda: torch.ops.fewbit.gelu(x, bi, l)))
print('fewbit.functional.gelu (no grad)   %.1f us' % host(lambda: F.gelu(x, bits=3)))
print('fewbit.GELU module (no grad)       %.1f us' % host(lambda: act(x)))
print('fewbit.ReLU module (no grad)       %.1f us' % host(lambda: relu(x)))
def fb():
    y = act(xr * 1.0); y.backward(g)
def vb():
    y = van(xr * 1.0); y.backward(g)
print('vanilla mul+GELU fwd+bwd           %.1f us' % host(vb, 500))
print('fewbit  mul+GELU fwd+bwd           %.1f us' % host(fb, 500))
# headline size through the module
X = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
G = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
def big(a):
    def f():
        xin = X.detach().requires_grad_(True)
        y = a(xin.view_as(xin) if False else xin.clone())
        y.backward(G)
    return f
for name, a in (('vanilla', van), ('fewbit', act)):
    f = big(a)
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): f()
    torch.cuda.synchronize(); print('%s clone+GELU fwd+bwd 4096^2: %.1f us' % (name, (time.perf_counter() - t0) / 200 * 1e6))
