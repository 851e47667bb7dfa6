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
# the raw operator against torch.nn.functional.gelu, per autograd route of the operator library (torch_ops.cpp header):
# forward with a node, and forward+backward, on a tensor small enough that only host time counts
import fewbit_amd
import torch.nn.functional as TF
op = torch.ops.fewbit.gelu.default
def raw_fwd():
    return op(xg.clone(), bi, l)
def raw_fb():
    op(xr * 1.0, bi, l).backward(g)
def van_fwd():
    return TF.gelu(xg.clone())
def van_fb():
    TF.gelu(xr * 1.0).backward(g)
def best(f, iters, rounds=5):
    return min(host(f, iters) for _ in range(rounds))
show('F.gelu(clone) fwd, input requires grad', best(van_fwd, 2000))
show('F.gelu(mul) fwd+bwd', best(van_fb, 1000))
if fewbit_amd.autograd_internals():
    for direct in (True, False):
        prev = fewbit_amd.autograd_route('direct_node', direct)
        tag = 'direct node' if direct else 'autograd::Function'
        show('torch.ops.fewbit.gelu(clone) fwd, input requires grad [%s]' % tag, best(raw_fwd, 2000))
        show('torch.ops.fewbit.gelu(mul) fwd+bwd [%s]' % tag, best(raw_fb, 1000))
        fewbit_amd.autograd_route('direct_node', prev)
else:
    show('torch.ops.fewbit.gelu(clone) fwd, input requires grad [autograd::Function]', best(raw_fwd, 2000))
    show('torch.ops.fewbit.gelu(mul) fwd+bwd [autograd::Function]', best(raw_fb, 1000))
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
