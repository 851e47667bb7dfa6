import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from fewbit_amd import cabi
from tests.helpers import from_raw
z = np.load('tests/golden/quantize_ref.npz')
dev = 'cuda'; n = 4096 * 4096
b = from_raw(z['gelu03_bf16_borders'], torch.bfloat16).to(dev); l = from_raw(z['gelu03_bf16_levels'], torch.bfloat16).to(dev)
x = torch.randn(n, device=dev).to(torch.bfloat16); gy = torch.randn(n, device=dev).to(torch.bfloat16)
y = torch.empty_like(x); gx = torch.empty_like(x)
st = torch.empty(cabi.state_nbytes(n, 3), dtype=torch.uint8, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
for _ in range(5):
    cabi.quantize_forward('gelu', x, b, out=y, state=st); cabi.quantize_backward(gy, st, l, out=gx)
torch.cuda.synchronize()
if which == 'fwd': cabi.quantize_forward('gelu', x, b, out=y, state=st)
else: cabi.quantize_backward(gy, st, l, out=gx)
torch.cuda.synchronize()
L = cabi.lib()
buf = np.zeros(8192 * 16, dtype=np.uint64)
L.fewbit_hip_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.fewbit_hip_debug_trace(buf.ctypes.data, buf.size) == 0
t = buf.reshape(8192, 16).astype(np.int64)
nz = t[:, 0] > 0
t = t[nz]; t0 = t[:, 0].min()
rel = (t - t0) * 0.01  # us
rel[t == 0] = np.nan
print(which, 'waves traced', nz.sum())
for slot in range(6):
    c = rel[:, slot]
    if np.all(np.isnan(c)): continue
    print(f'slot {slot}: min {np.nanmin(c):6.2f}  p10 {np.nanpercentile(c,10):6.2f}  med {np.nanmedian(c):6.2f}  p90 {np.nanpercentile(c,90):6.2f}  max {np.nanmax(c):6.2f} us')
# per-wave durations
d = rel[:, 1:4] - rel[:, 0:3]
print('stage durations median (start->init, init->proc1 done, proc1->proc2 done):', np.nanmedian(d, axis=0))
# who is late? (slot 1 = first data + tables ready)
late = rel[:, 1] > np.nanpercentile(rel[:, 1], 85)
wid = np.nonzero(nz)[0]
blk = wid // 4 if which != 'lut' else wid // 16
print('late waves:', late.sum(), ' distinct blocks containing late waves:', len(set(blk[late])), 'of', len(set(blk)))
xcd = blk % 8
print('late fraction by XCD (block%8):', [round(float(late[xcd == k].mean()), 2) for k in range(8)])
print('late fraction by wave-in-block:', [round(float(late[(wid % 4) == k].mean()), 2) for k in range(4)])
# are whole blocks late together?
import collections
cnt = collections.Counter(blk[late])
print('late waves per late block histogram:', sorted(collections.Counter(cnt.values()).items()))
order = np.argsort(wid)
seg = late[order].astype(int)
runs = np.diff(np.flatnonzero(np.diff(np.concatenate([[0], seg, [0]]))))[::2]
print('run lengths of consecutive late wave ids: max', runs.max() if len(runs) else 0, 'median', np.median(runs) if len(runs) else 0)
print('first 40 late wave ids:', wid[late][:40].tolist())
