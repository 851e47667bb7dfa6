"""Host-side cost per call of the op-level path (module -> functional -> torch.ops -> autograd -> C-ABI) for tensors
small enough that the GPU is never the bottleneck, and the op-level fwd+bwd time at the headline size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import json, torch, fewbit
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
res = {}
def show(label, us):
    res[label] = round(us, 2); print('%-44s %.1f us' % (label, us))
show('torch.nn.GELU fwd (input without grad)', host(lambda: van(x)))
show('torch.ops.fewbit.gelu (input without grad)', host(lambda: torch.ops.fewbit.gelu(x, bi, l)))
show('fewbit.functional.gelu (input without grad)', host(lambda: F.gelu(x, bits=3)))
show('fewbit.GELU module (input without grad)', host(lambda: act(x)))
show('fewbit.ReLU module (input without grad)', host(lambda: relu(x)))
xg = (x.clone().requires_grad_(True) * 1.0)          # non-leaf that requires grad: the call builds an autograd node
show('torch.nn.GELU fwd (input requires grad)', host(lambda: van(xg)))
show('fewbit.GELU module (input requires grad)', host(lambda: act(xg)))
def fb():
    y = act(xr * 1.0); y.backward(g)
def vb():
    y = van(xr * 1.0); y.backward(g)
show('vanilla mul+GELU fwd+bwd', host(vb, 500))
show('fewbit  mul+GELU fwd+bwd', host(fb, 500))
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
    torch.cuda.synchronize(); show('%s clone+GELU fwd+bwd 4096^2 bf16 (eager)' % name, (time.perf_counter() - t0) / 200 * 1e6)
os.makedirs('gpurun_out', exist_ok=True)
json.dump({'what': 'host wall time per call, 1024-element bf16 tensor unless stated (GPU never the bottleneck), scratch/hostcost.py',
           'us': res}, open('gpurun_out/hostcost.json', 'w'), indent=1)
