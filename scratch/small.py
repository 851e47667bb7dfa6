"""small-tensor launches for a rocprofv3 kernel trace: BASELINE config 0 (relu 1 bit 1024x1024 fp32) and friends"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev='cuda'
for dtype, n in ((torch.float32, 1024*1024), (torch.bfloat16, 1024*1024), (torch.bfloat16, 256*1024)):
    x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); gy = torch.randn(n, device=dev).to(dtype); gx = torch.empty_like(x)
    st = torch.empty(cabi.state_nbytes(n, 4), dtype=torch.uint8, device=dev)
    b, l = store.get('gelu', 3, dev, dtype); b = b[1:-1].contiguous()
    for _ in range(50):
        cabi.stepwise1_forward('relu', x, out=y, state=st); cabi.stepwise1_backward('relu', gy, st, out=gx)
        cabi.quantize_forward('gelu', x, b, out=y, state=st); cabi.quantize_backward(gy, st, l, out=gx)
        torch.relu(x); torch.nn.functional.gelu(x)
    torch.cuda.synchronize()
