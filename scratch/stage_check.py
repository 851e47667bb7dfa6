"""state-staging experiment (FEWBIT_STATE_STAGE build): same bytes as the production library for every chunk length"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store
stage = ctypes.CDLL(os.path.abspath('scratch/libfewbit_hip_stage.so'))
vp, sz, i32, dbl = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double
stage.fewbit_hip_quantize_forward.argtypes = [i32, i32, vp, vp, vp, sz, vp, i32, dbl, dbl, vp]
stage.fewbit_hip_tune.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
bo, lv = store.get('gelu', 3, 'cuda', torch.bfloat16); bo = bo[1:-1].contiguous()
ok = True
for n in (7 * 1024 * 1024 + 3, 16 * 1024 * 1024, 512 * 16 * 5 * 3 + 512 * 7 + 5):
    x = (torch.randn(n, device='cuda') * 1.5).to(torch.bfloat16)
    y0, s0 = cabi.quantize_forward('gelu', x, bo)
    for chunk in (1, 2, 3, 4, 5, 7, 8, 13):
        stage.fewbit_hip_tune(b'lut_min', 0); stage.fewbit_hip_tune(b'lut_chunk', chunk)
        y = torch.empty_like(x); st = torch.full_like(s0, 0xAA)
        rc = stage.fewbit_hip_quantize_forward(2, 2, x.data_ptr(), y.data_ptr(), st.data_ptr(), n, bo.data_ptr(), bo.numel(), 0.0, 0.0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        same = rc == 0 and torch.equal(st, s0) and torch.equal(y.view(torch.int16), y0.view(torch.int16))
        ok &= same
        print(n, chunk, 'same' if same else 'DIFFERENT', int((st != s0).sum()))
print('ALL SAME' if ok else 'MISMATCH')
