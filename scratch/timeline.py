"""step time as a function of time since the first launch: one event every CH steps over a long run (DVFS / power ramp?)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import time, torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev = 'cuda'
n = 4096 * 4096; dtype = torch.bfloat16; k = 3
bo, lv = store.get('gelu', k, dev, dtype); bo = bo[1:-1].contiguous()
x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); gy = torch.randn(n, device=dev).to(dtype); gx = torch.empty_like(x)
st = torch.empty(cabi.state_nbytes(n, k), dtype=torch.uint8, device=dev)
f = cabi.bind_forward('gelu', x, bo, out=y, state=st); b = cabi.bind_backward(gy, st, lv, out=gx)
torch.cuda.synchronize()
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
time.sleep(idle)                      # GPU idle before the run, like a fresh process
CH, NCH = 10, 300
ev = [torch.cuda.Event(enable_timing=True) for _ in range(NCH + 1)]
ev[0].record()
for i in range(NCH):
    for _ in range(CH): f(); b()
    ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) * 1000 / CH for i in range(NCH)]
print('idle %.1fs; us/step by chunk of %d steps:' % (idle, CH))
print(' first 30 :', ' '.join('%.1f' % v for v in t[:30]))
print(' 30..100  :', ' '.join('%.1f' % v for v in t[30:100:5]))
print(' 100..300 :', ' '.join('%.1f' % v for v in t[100:300:10]))
