"""per-wave timeline of one forward (pattern table) / backward launch at the headline size (-DFEWBIT_TRACE build)"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev = 'cuda'; n = 4096 * 4096
bo, lv = store.get('gelu', 3, dev, torch.bfloat16); bo = bo[1:-1].contiguous()
x = torch.randn(n, device=dev).to(torch.bfloat16); gy = torch.randn(n, device=dev).to(torch.bfloat16)
y = torch.empty_like(x); gx = torch.empty_like(x)
st = torch.empty(cabi.state_nbytes(n, 3), dtype=torch.uint8, device=dev)
L = cabi.lib()
L.fewbit_hip_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for which in ('fwd', 'bwd'):
    for _ in range(3000):
        cabi.quantize_forward('gelu', x, bo, out=y, state=st); cabi.quantize_backward(gy, st, lv, out=gx)
    torch.cuda.synchronize()
    if which == 'fwd':
        cabi.quantize_backward(gy, st, lv, out=gx); cabi.quantize_forward('gelu', x, bo, out=y, state=st)
    else:
        cabi.quantize_forward('gelu', x, bo, out=y, state=st); cabi.quantize_backward(gy, st, lv, out=gx)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    assert L.fewbit_hip_debug_trace(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(8192, 16).astype(np.int64)
    nz = t[:, 0] > 0
    t = t[nz]; t0 = t[:, 0].min()
    rel = (t - t0) * 0.01
    rel[t == 0] = np.nan
    print(which, 'waves traced', int(nz.sum()))
    names = ['start', 'init done', 'tile1 done', 'tile2 done', 'tile3 done', 'tile4 done']
    for slot in range(6):
        c = rel[:, slot]
        if np.all(np.isnan(c)): continue
        print(f'  {names[slot]:10s}: min {np.nanmin(c):6.2f}  p10 {np.nanpercentile(c,10):6.2f}  med {np.nanmedian(c):6.2f}  p90 {np.nanpercentile(c,90):6.2f}  max {np.nanmax(c):6.2f} us')
    last = np.nanmax(rel, axis=1)
    wid = np.nonzero(nz)[0]
    wpb = 16 if which == 'fwd' else 4
    xcd = (wid // wpb) % 8
    print('  finish time by XCD (median / max):', [f'{np.median(last[xcd==k]):.2f}/{last[xcd==k].max():.2f}' for k in range(8)])
    print('  start  time by XCD (median / max):', [f'{np.median(rel[xcd==k,0]):.2f}/{rel[xcd==k,0].max():.2f}' for k in range(8)])
    st = rel[:, 0]
    print('  fraction of waves starting after 1 / 2 / 4 us:', [round(float((st > v).mean()), 4) for v in (1, 2, 4)])
    blk = wid // wpb
    lateb = sorted(set(blk[st > 2.0].tolist()))
    print('  blocks with a wave starting after 2 us:', len(lateb), 'of', len(set(blk.tolist())), ' first ids', lateb[:24], ' last ids', lateb[-8:])
