"""The in-place bf16 forward inside a training step against x.neg_() in the same state, under rocprofv3 --pmc:
the states of scratch/insitu_ab.py -- back to back / right behind the GEMM that produced x / behind that GEMM plus 100 MB of
other traffic -- run one after the other with a fixed number of dispatches each, so that the counter rows of a pass can be
attributed by kernel name and dispatch order (tools/profile_insitu_pmc.sh parses them with the manifest this prints)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store

dev = 'cuda'
rows, din, dout = 16384, 768, 3072
REPS = int(os.environ.get('REPS', '30'))
bo, lv = store.get('gelu', 3, dev, torch.bfloat16); bo = bo[1:-1].contiguous()
h = torch.randn(rows, din, device=dev).to(torch.bfloat16)
w = (torch.randn(dout, din, device=dev) * 0.05).to(torch.bfloat16)
x = torch.empty(rows, dout, device=dev, dtype=torch.bfloat16)
y = torch.empty_like(x)
state = torch.empty(cabi.state_nbytes(x.numel(), 3), dtype=torch.uint8, device=dev)
other = torch.randn(50 * 2**20, device=dev).to(torch.bfloat16)
other2 = torch.empty_like(other)

kernels = {'fwd_in_place': lambda: cabi.quantize_forward('gelu', x, bo, out=x, state=state),
           'fwd_out_of_place': lambda: cabi.quantize_forward('gelu', x, bo, out=y, state=state),
           'neg_in_place': lambda: x.neg_()}


def gemm():
    torch.matmul(h, w.t(), out=x)


def gemm_and_traffic():
    torch.matmul(h, w.t(), out=x)
    other2.copy_(other)


states = {'back_to_back': lambda: None, 'after_gemm': gemm, 'after_gemm_and_100MB': gemm_and_traffic}
gemm(); torch.cuda.synchronize()
manifest = []
for kname, kernel in kernels.items():
    for sname, before in states.items():
        for _ in range(REPS):
            before()
            kernel()
        torch.cuda.synchronize()
        manifest.append({'kernel': kname, 'state': sname, 'dispatches': REPS})
print('MANIFEST ' + json.dumps(manifest), flush=True)
