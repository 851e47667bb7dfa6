"""RandomizedLinear (16384 rows, 768 -> 3072, ratio 0.2): fwd+bwd time by sketch kind, beside nn.Linear"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch, fewbit
dev = 'cuda'
def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
for dtype in (torch.bfloat16, torch.float32):
    x = torch.randn(16384, 768, device=dev, dtype=dtype, requires_grad=True)
    g = torch.randn(16384, 3072, device=dev, dtype=dtype)
    lin = torch.nn.Linear(768, 3072, device=dev, dtype=dtype)
    def step(m):
        def f():
            m.zero_grad(set_to_none=True); x.grad = None
            m(x).backward(g)
        return f
    print(str(dtype)[6:], 'nn.Linear %.2f ms' % timeit(step(lin)), flush=True)
    for kind in ('gaussian', 'rademacher', 'dct', 'dft'):
        m = fewbit.RandomizedLinear(768, 3072, device=dev, dtype=dtype, proj_dim_ratio=0.2, matmul=kind)
        print(str(dtype)[6:], kind, '%.2f ms' % timeit(step(m)), flush=True)
    p = int(0.2 * 16384)
    print('   randn(p,B) %.2f ms | randint %.2f ms | S@X %.2f ms' % (
        timeit(lambda: torch.randn(p, 16384, device=dev, dtype=dtype)),
        timeit(lambda: torch.randint(0, 2, (p, 16384), device=dev, dtype=torch.int8)),
        timeit(lambda s=torch.randn(p, 16384, device=dev, dtype=dtype): s @ x.detach())))
