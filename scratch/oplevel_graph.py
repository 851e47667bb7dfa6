"""operator-level fwd+bwd (module -> torch.ops.fewbit -> autograd) at the headline size, eager and as a hipGraph replay"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch, fewbit
dev = 'cuda'
X = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16, requires_grad=True)
G = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
def run(act):
    y = act(X.clone())
    return torch.autograd.grad(y, X, G)[0]
for name, act in (('vanilla nn.GELU', torch.nn.GELU()), ('fewbit.GELU(bits=3)', fewbit.GELU(bits=3))):
    for _ in range(10): run(act)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): run(act)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 300 * 1e6
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): run(act)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run(act)
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 500 * 1e6
    print(f'{name}: clone + forward + backward at 4096x4096 bf16: eager {eager:.1f} us, hipGraph replay {graph:.1f} us', flush=True)
