import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch, fewbit
dev='cuda'
for dtype in (torch.float32, torch.bfloat16):
    lin = torch.nn.Linear(768, 3072).to(dev, dtype)
    x = torch.randn(16384, 768, device=dev, dtype=dtype, requires_grad=True)
    for name, act in (('vanilla', torch.nn.GELU()), ('fewbit', fewbit.GELU(bits=3))):
        def step():
            y = act(lin(x))
            y.backward(torch.ones_like(y)) if False else y.sum().backward()
        for _ in range(5): step()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
        print(dtype, name, 'linear+act fwd+bwd %.3f ms'%(dt*1e3))
    from torch.profiler import profile, ProfilerActivity
    act = fewbit.GELU(bits=3)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            y = act(lin(x)); y.sum().backward()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=12, max_name_column_width=60))
