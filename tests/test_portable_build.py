"""The operator library restricted to libtorch's PUBLIC API (-DFEWBIT_AUTOGRAD_INTERNALS=0: what a build against any torch
release other than the verified one gets, fewbit_amd/csrc/torch_ops.cpp header): it must keep compiling, and give the same
bytes and gradients.  Built on demand into scratch/ (make -C fewbit_amd/csrc portable, ~45 s) and driven in a fresh process
through FEWBIT_OPS_LIB, because one process can register the `fewbit` operator namespace only once."""
import os
import subprocess
import sys
import textwrap

import pytest

from helpers import ROOT

SCRIPT = textwrap.dedent('''
    import torch, pytest
    import fewbit_amd, oracle
    from fewbit_amd.store import store
    assert fewbit_amd.native_loaded(), fewbit_amd.native_error()
    assert not fewbit_amd.autograd_internals()
    assert fewbit_amd.autograd_route('direct_node') is False and fewbit_amd.autograd_route('base_dirty') is False
    try:
        fewbit_amd.autograd_route('direct_node', True)
        raise SystemExit('enabling an internal route on the public-API build must fail')
    except RuntimeError:
        pass
    g = torch.Generator().manual_seed(0)
    x, gy = torch.randn(1003, generator=g) * 2, torch.randn(1003, generator=g)
    for name, bits in (('gelu', 3), ('silu', 2)):
        inner, levels = store.get_inner(name, bits, torch.device('cpu'), torch.float32)
        _, state_o, _ = oracle.quantize(name, x, inner)
        xx = x.clone().requires_grad_()
        saved = []
        with torch.autograd.graph.saved_tensors_hooks(lambda t: (saved.append(t), t)[1], lambda t: t):
            inp = xx * 1.0
            y = getattr(torch.ops.fewbit, name)(inp, inner, levels)
        assert 'FewbitPackedBackward' not in y.grad_fn.name()
        assert (y.data_ptr() == inp.data_ptr()) == (name != 'gelu')
        assert torch.equal([t for t in saved if t.dtype == torch.uint8][0], state_o)
        y.backward(gy)
        assert torch.equal(xx.grad, oracle.quantize_backward(gy, state_o, levels))
    v = (x.clone().requires_grad_() * 1.0).view(17, 59)               # a whole-tensor view: the general in-place route
    out = torch.ops.fewbit.relu(v)
    out.backward(gy.view(17, 59))
    print('portable ok')
''')


def test_public_api_only_build_compiles_and_agrees():
    r = subprocess.run(['make', '-C', str(ROOT / 'fewbit_amd' / 'csrc'), 'portable'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, FEWBIT_OPS_LIB=str(ROOT / 'scratch' / 'libfewbit_portable.so'), PYTHONPATH=str(ROOT))
    r = subprocess.run([sys.executable, '-c', SCRIPT], env=env, capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert r.returncode == 0 and 'portable ok' in r.stdout, r.stderr[-3000:]


GPU_SCRIPT = textwrap.dedent('''
    import torch
    import fewbit_amd, oracle
    from fewbit_amd.store import store
    assert fewbit_amd.native_loaded(), fewbit_amd.native_error()
    assert not fewbit_amd.autograd_internals()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        x, gy = (torch.randn(70001, generator=g) * 2).to(dtype), torch.randn(70001, generator=g).to(dtype)
        for name, bits in (('gelu', 3), ('silu', 4)):
            inner, levels = store.get_inner(name, bits, torch.device('cpu'), dtype)
            _, state_o, _ = oracle.quantize(name, x, inner)
            gx_o = oracle.quantize_backward(gy, state_o, levels)
            xx = x.to(dev).requires_grad_()
            saved = []
            with torch.autograd.graph.saved_tensors_hooks(lambda t: (saved.append(t), t)[1], lambda t: t):
                inp = xx.clone()
                y = getattr(torch.ops.fewbit, name)(inp, inner.to(dev), levels.to(dev))
            assert 'FewbitPackedBackward' not in y.grad_fn.name()           # the torch::autograd::Function route
            assert y.data_ptr() == inp.data_ptr()                            # in place on the device
            state = [t for t in saved if t.dtype == torch.uint8][0]
            assert torch.equal(state.cpu()[:state_o.numel()], state_o) and int(state.cpu()[state_o.numel():].sum()) == 0
            y.backward(gy.to(dev))
            assert torch.equal(xx.grad.cpu().view(torch.int16 if dtype != torch.float32 else torch.int32),
                               gx_o.view(torch.int16 if dtype != torch.float32 else torch.int32)), (name, dtype)
        # the 1-bit family and the reference caller's route (in place on the 3-D view of a linear layer's output)
        xx = x.to(dev).requires_grad_()
        out = torch.ops.fewbit.relu(xx.clone())
        out.backward(gy.to(dev))
        assert torch.equal(xx.grad.cpu(), torch.where(x > 0, gy, torch.zeros_like(gy)))
    lin = torch.nn.Linear(64, 128).to(dev)
    h = lin(torch.randn(4, 8, 64, device=dev))
    inner, levels = store.get_inner('gelu', 3, dev, torch.float32)
    want = torch.nn.functional.gelu(h.detach())
    out = torch.ops.fewbit.gelu(h, inner, levels)
    assert torch.allclose(out, want, atol=1e-6)
    out.sum().backward()
    assert lin.weight.grad is not None and bool(torch.isfinite(lin.weight.grad).all())
    torch.cuda.synchronize()
    print('portable gpu ok')
''')


@pytest.mark.gpu
def test_public_api_only_build_on_the_gpu():
    """the route every torch release other than the verified one gets (torch::autograd::Function, no autograd internals;
    cf. the reference's ContinousCudaFunction, fewbit/cuda/activation.cc:345-381), run on cuda:0: packed state bytes and gradients
    against the oracle for three dtypes, in place, and the reference caller's 3-D view route"""
    r = subprocess.run(['make', '-C', str(ROOT / 'fewbit_amd' / 'csrc'), 'portable'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, FEWBIT_OPS_LIB=str(ROOT / 'scratch' / 'libfewbit_portable.so'), PYTHONPATH=str(ROOT))
    r = subprocess.run([sys.executable, '-c', GPU_SCRIPT], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0 and 'portable gpu ok' in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
