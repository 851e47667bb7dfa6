"""The operator library on HOST tensors (dispatch keys AutogradCPU / CPU of fewbit_amd/csrc/torch_ops.cpp), no GPU.

The reference registers `gelu` under the CPU key and exposes `quantize` / `quantize_backward` for host tensors
(fewbit/cpu/gelu.cc:74-76, fewbit/fewbit.cc:6-7); here every operator has a host implementation with the same packed
state as the GPU kernels.  Checked bit for bit against the outputs of the reference itself (tests/golden/quantize_ref.npz,
codec_ref.npz), and against the oracle for what the reference cannot run."""
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import fewbit
import oracle
from fewbit_amd.store import store
from helpers import DTYPES, GOLDEN, assert_bit_equal, from_raw, full_size_inputs, load_tables, sha256_of


@pytest.fixture(scope='module')
def qref():
    with np.load(GOLDEN / 'quantize_ref.npz') as z:
        return {k: z[k].copy() for k in z.files}


def _cases(qref):
    return sorted({k.rsplit('_', 1)[0] for k in qref if k.endswith('_x')})


def test_quantize_ops_match_reference_run_bitwise(qref):
    """torch.ops.fewbit.quantize / quantize_backward / gelu on host tensors == the reference's own outputs."""
    seen = 0
    with warnings.catch_warnings():
        warnings.simplefilter('error')                        # e.g. "autograd kernel was not registered"
        for key in _cases(qref):
            name_k, dt, _ = key.split('_')
            dtype = DTYPES[dt]
            x, gy = from_raw(qref[key + '_x'], dtype), from_raw(qref[key + '_gy'], dtype)
            b, l = from_raw(qref[f'{name_k}_{dt}_borders'], dtype), from_raw(qref[f'{name_k}_{dt}_levels'], dtype)
            k = int(name_k[-2:])
            y, state = torch.ops.fewbit.quantize(x, b)
            ref_state = torch.from_numpy(qref[key + '_state'])            # reference length: ceil(k*n/8)
            assert state.numel() == k * ((x.numel() + 7) // 8)             # here: k*ceil(n/8), zero padded
            assert torch.equal(state[:ref_state.numel()], ref_state), key
            assert int(state[ref_state.numel():].sum()) == 0
            want_gx = from_raw(qref[key + '_gx'], dtype)
            assert_bit_equal(torch.ops.fewbit.quantize_backward(gy, state, l), want_gx, key)
            assert_bit_equal(torch.ops.fewbit.quantize_backward(gy, ref_state, l), want_gx, key + ' (reference-sized state)')
            if name_k.startswith('gelu'):
                assert_bit_equal(y, from_raw(qref[key + '_y'], dtype), key + ' y')
                xx = x.clone().requires_grad_()
                inp = xx.clone()
                out = torch.ops.fewbit.gelu(inp, b, l)
                # host tensors: a fresh result, the input stays intact -- what the reference's CPU operator does
                # (fewbit/cpu/gelu.cc:7-31,47-56) whatever the schema's Tensor(a!) says
                assert out.data_ptr() != inp.data_ptr()
                assert_bit_equal(inp.detach(), x, key + ' input left intact')
                out.backward(gy)
                assert_bit_equal(xx.grad, want_gx, key + ' autograd')
            seen += 1
    assert seen >= 70


@pytest.mark.parametrize('name', [n for n in fewbit.functional.CONTINOUS])
@pytest.mark.parametrize('dtype', (torch.float32, torch.bfloat16, torch.float16))
def test_every_continuous_op_on_host_matches_oracle_state(name, dtype):
    g = torch.Generator().manual_seed(11)
    n = 1003
    x = (torch.randn(n, generator=g) * 2).to(dtype)
    x[:5] = torch.tensor([float('nan'), float('inf'), -float('inf'), 0.0, -0.0]).to(dtype)
    gy = torch.randn(n, generator=g).to(dtype)
    for bits in (1, 2, 3, 4):
        inner, levels = store.get_inner(name, bits, torch.device('cpu'), dtype)
        _, state_o, _ = oracle.quantize(name, x, inner)
        gx_o = oracle.quantize_backward(gy, state_o, levels)
        xx = x.clone().requires_grad_()
        saved = []
        with torch.autograd.graph.saved_tensors_hooks(lambda t: (saved.append(t), t)[1], lambda t: t):
            y = getattr(torch.ops.fewbit, name)(xx.clone(), inner, levels)
        packed = [t for t in saved if t.dtype == torch.uint8]
        assert len(packed) == 1 and torch.equal(packed[0], state_o)        # the packed state is what is saved
        y.backward(gy)
        assert_bit_equal(xx.grad, gx_o, f'{name} k={bits}')
        ref = getattr(F, name)(x) if hasattr(F, name) else getattr(torch, name)(x)
        assert_bit_equal(y.detach(), ref, f'{name} forward')


@pytest.mark.parametrize('name,args', [('hardshrink', (0.5,)), ('hardsigmoid', ()), ('hardtanh', (-1.0, 1.0)),
                                       ('leaky_relu', (0.01,)), ('relu', ()), ('relu6', ()), ('softshrink', (0.5,)),
                                       ('threshold', (0.25, -3.0))])
@pytest.mark.parametrize('dtype', (torch.float32, torch.bfloat16))
def test_stepwise1_ops_on_host(name, args, dtype):
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(777, generator=g) * 3).to(dtype)
    x[:6] = torch.tensor([0.0, -0.0, 3.0, -3.0, 6.0, 0.5]).to(dtype)
    gy = torch.randn(777, generator=g).to(dtype)
    y_o, state_o = oracle.stepwise1_forward(name, x, *args)
    gx_o = oracle.stepwise1_backward(name, gy, state_o, *(args[:1] if name == 'leaky_relu' else ()))
    xx = x.clone().requires_grad_()
    saved = []
    with torch.autograd.graph.saved_tensors_hooks(lambda t: (saved.append(t), t)[1], lambda t: t):
        y = getattr(torch.ops.fewbit, name)(xx.clone(), *args)
    assert len(saved) == 1 and torch.equal(saved[0], state_o)
    y.backward(gy)
    # values are ATen's on the host (sign of a zero result included, where ATen's relu keeps -0 and the kernels'
    # `x <= 0 -> 0` rule, fewbit/cuda/codec.cu:412-425, gives +0); equal as numbers to the oracle's
    assert_bit_equal(y.detach(), getattr(F, name)(x, *args), name)
    assert torch.equal(y.detach().float(), y_o.float())
    assert_bit_equal(xx.grad, gx_o, name)


def test_relu_1bit_matches_reference_packed_bits():
    with np.load(GOLDEN / 'codec_ref.npz') as z:
        for dt, dtype in DTYPES.items():
            x = from_raw(z[f'relu01_{dt}_x'], dtype)
            saved = []
            with torch.autograd.graph.saved_tensors_hooks(lambda t: (saved.append(t), t)[1], lambda t: t):
                y = torch.ops.fewbit.relu(x.clone().requires_grad_().clone())
            assert torch.equal(saved[0], torch.from_numpy(z[f'relu01_{dt}_state']))
            assert torch.equal(y.detach().float().nan_to_num(7.0), from_raw(z[f'relu01_{dt}_y'], dtype).float().nan_to_num(7.0))


def test_host_state_layout_is_the_device_layout():
    """ragged sizes, non power-of-two tables, wide tables: the host pack equals the oracle's (== reference Deflate)"""
    g = torch.Generator().manual_seed(3)
    for nlev in (2, 3, 5, 8, 9, 17, 100, 256):
        borders = torch.sort(torch.randn(nlev - 1, generator=g))[0]
        levels = torch.randn(nlev, generator=g)
        for n in (1, 7, 8, 9, 63, 64, 65, 1001):
            x = torch.randn(n, generator=g)
            gy = torch.randn(n, generator=g)
            _, state_o, k = oracle.quantize('identity', x, borders)
            xx = x.clone().requires_grad_()
            saved = []
            with torch.autograd.graph.saved_tensors_hooks(lambda t: (saved.append(t), t)[1], lambda t: t):
                y = torch.ops.fewbit.stepwise(xx.clone(), borders, levels)
            st = [t for t in saved if t.dtype == torch.uint8][0]
            assert torch.equal(st, state_o), (nlev, n)
            y.backward(gy)
            assert_bit_equal(xx.grad, oracle.quantize_backward(gy, state_o, levels))


def test_host_ops_without_autograd_and_errors():
    inner, levels = store.get_inner('silu', 2, torch.device('cpu'), torch.float32)
    x = torch.randn(100)
    with torch.inference_mode():
        y = torch.ops.fewbit.silu(x.clone(), inner, levels)
    assert torch.equal(y, F.silu(x))
    with torch.no_grad():
        assert torch.equal(torch.ops.fewbit.relu(x.clone()), F.relu(x))
    with pytest.raises(RuntimeError):
        torch.ops.fewbit.gelu(x.clone(), inner, levels[:-1])              # table sizes
    with pytest.raises(RuntimeError):
        torch.ops.fewbit.gelu(x.clone(), inner.double(), levels.double())  # dtype mismatch
    with pytest.raises(RuntimeError):
        torch.ops.fewbit.quantize_backward(x, torch.zeros(3, dtype=torch.uint8), levels)   # state too small
    with torch.inference_mode(), pytest.raises(RuntimeError):
        torch.ops.fewbit.gelu(x.clone(), inner.double(), levels.double())  # the same check without an autograd node
    # host tensors of any layout are accepted, like the reference's CPU operator: logical (row-major) element order
    xt = torch.randn(6, 10).t()
    xg = xt.clone().requires_grad_()                                      # (clone keeps the transposed strides)
    with pytest.raises(RuntimeError, match='leaf Variable'):               # in place on a leaf that requires grad: autograd's own error
        torch.ops.fewbit.silu(xg, inner, levels)
    xin = xg * 1.0                                                         # (element-wise ops keep the strides too)
    assert not xin.is_contiguous()
    y = torch.ops.fewbit.silu(xin, inner, levels)
    assert y.data_ptr() == xin.data_ptr() and not y.is_contiguous()       # Tensor(a!): written back through the strides
    assert torch.equal(y, F.silu(xt.contiguous()))          # (ATen's strided and contiguous silu differ by an ulp)
    y.sum().backward()
    assert torch.equal(xg.grad, levels[torch.searchsorted(inner, xt.contiguous())])
    _, state_t = torch.ops.fewbit.quantize(xt, inner)
    _, state_c = torch.ops.fewbit.quantize(xt.contiguous(), inner)
    assert torch.equal(state_t, state_c)


def test_full_size_c2_digest_of_the_reference_run_on_the_host():
    """BASELINE configs[1] at full size (4096x4096 bf16): the host operators and the oracle both reproduce the SHA-256
    digests of what the reference itself computed (tests/golden/fullsize_digests.json)."""
    import json
    case = 'c2_gelu3_bf16_4096x4096'
    want = json.loads((GOLDEN / 'fullsize_digests.json').read_text())['cases'][case]
    x, gy, inner, levels = full_size_inputs(case, load_tables())
    if sha256_of(x) != want['x'] or sha256_of(gy) != want['gy']:
        pytest.skip('LOUD SKIP: seeded host inputs differ from the build container; digests cannot be compared here')
    _, state = torch.ops.fewbit.quantize(x, inner)
    assert sha256_of(state) == want['state']
    assert sha256_of(torch.ops.fewbit.quantize_backward(gy, state, levels)) == want['gx']
    _, state_o, _ = oracle.quantize('gelu', x, inner)
    assert sha256_of(state_o) == want['state']
    assert sha256_of(oracle.quantize_backward(gy, state_o, levels)) == want['gx']


@pytest.mark.parametrize('dtype', (torch.bfloat16, torch.float16))
def test_host_ops_every_16bit_pattern(dtype):
    """All 65 536 patterns of a 16-bit dtype through the host operators (gelu k = 1..4, a custom table with borders on
    special values): packed state and gradient bit-identical to the oracle."""
    x = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(dtype)
    gy = torch.full((65536,), 1.5).to(dtype)
    for bits in (1, 2, 3, 4):
        inner, levels = store.get_inner('gelu', bits, torch.device('cpu'), dtype)
        _, state = torch.ops.fewbit.quantize(x, inner)
        _, state_o, _ = oracle.quantize('gelu', x, inner)
        assert torch.equal(state, state_o), bits
        assert_bit_equal(torch.ops.fewbit.quantize_backward(gy, state, levels), oracle.quantize_backward(gy, state_o, levels))
    edge = torch.tensor([-float('inf'), -1.0, -0.0, 6e-8, 1.0, float('inf')]).to(dtype)
    levels = torch.arange(7.0).to(dtype)
    xx = x.clone().requires_grad_()
    saved = []
    with torch.autograd.graph.saved_tensors_hooks(lambda t: (saved.append(t), t)[1], lambda t: t):
        torch.ops.fewbit.stepwise(xx.clone(), edge, levels)
    _, state_o, _ = oracle.quantize('identity', x, edge)
    assert torch.equal([t for t in saved if t.dtype == torch.uint8][0], state_o)


# ---- the in-place rule on the host, and the two autograd routes ------------------------------------------------------
IN_PLACE_OPS = [(n, 'table') for n in fewbit.functional.CONTINOUS if n != 'gelu'] + \
               [('stepwise', 'table'), ('relu', ()), ('relu6', ()), ('hardsigmoid', ()), ('hardshrink', (0.5,)), ('softshrink', (0.5,)),
                ('hardtanh', (-1.0, 1.0)), ('leaky_relu', (0.01,)), ('threshold', (0.25, -3.0))]


@pytest.mark.parametrize('name,args', IN_PLACE_OPS)
def test_host_operators_honour_their_in_place_schema_except_gelu(name, args):
    """`Tensor(a!) self -> Tensor(a!)` on host tensors: every operator writes its result into `self` and returns it, with and
    without an autograd node -- the same call has the same effect on its argument on the GPU and on the host.  The one
    exception is pinned too: `gelu`, which the reference implements for the host out of place (fewbit/cpu/gelu.cc:7-31)."""
    x = torch.randn(300) * 2
    if args == 'table':
        args = store.get_inner('silu' if name == 'stepwise' else name, 3, torch.device('cpu'), torch.float32)
    op = getattr(torch.ops.fewbit, name)
    for with_node in (True, False):
        inp = x.clone().requires_grad_(with_node) * 1.0 if with_node else x.clone()
        version = inp._version
        out = op(inp, *args)
        assert out.data_ptr() == inp.data_ptr() and inp._version > version, (name, with_node)
        assert torch.equal(inp.detach(), out.detach()) and out.requires_grad == with_node
        if name != 'stepwise':
            ref = getattr(F, name)(x, *(() if len(args) == 2 and isinstance(args[0], torch.Tensor) else args))
            assert torch.equal(out.detach(), ref), name
    # gelu: fresh tensor, input intact, in both flavours
    inner, levels = store.get_inner('gelu', 3, torch.device('cpu'), torch.float32)
    for with_node in (True, False):
        inp = x.clone().requires_grad_(with_node) * 1.0 if with_node else x.clone()
        out = torch.ops.fewbit.gelu(inp, inner, levels)
        assert out.data_ptr() != inp.data_ptr() and torch.equal(inp.detach(), x)


def test_out_of_place_variants_never_touch_their_input_on_the_host():
    inner, levels = store.get_inner('silu', 2, torch.device('cpu'), torch.float32)
    x = torch.randn(64)
    for inp in (x.clone(), x.clone().requires_grad_() * 1.0):
        y = torch.ops.fewbit.continuous_out(inp, inner, levels, 8, 0.0, 0.0)         # 8 = silu (include/fewbit_hip.h)
        z = torch.ops.fewbit.stepwise1_out(inp, 4, 0.0, 0.0)                         # 4 = relu
        assert y.data_ptr() != inp.data_ptr() and z.data_ptr() != inp.data_ptr() and torch.equal(inp.detach(), x)
        assert torch.equal(y.detach(), F.silu(x)) and torch.equal(z.detach(), F.relu(x))


def test_direct_node_and_function_routes_agree_on_the_host():
    """fewbit_torch_route('direct_node'): the hand-written backward node and the torch::autograd::Function fallback give the
    same values, the same saved bytes, the same gradients and the same errors (host tensors; tests/test_gpu_ops.py repeats
    this on the device)."""
    import fewbit_amd
    if not fewbit_amd.autograd_internals():
        pytest.skip('operator library built without the internal-API routes')
    inner, levels = store.get_inner('silu', 3, torch.device('cpu'), torch.float32)
    x, gy = torch.randn(999) * 2, torch.randn(999)
    results = {}
    prev = fewbit_amd.autograd_route('direct_node')
    try:
        for direct in (True, False):
            fewbit_amd.autograd_route('direct_node', direct)
            assert fewbit_amd.autograd_route('direct_node') is direct
            got = []
            for op, args in ((torch.ops.fewbit.silu, (inner, levels)), (torch.ops.fewbit.leaky_relu, (0.2,)),
                             (torch.ops.fewbit.gelu, store.get_inner('gelu', 2, torch.device('cpu'), torch.float32))):
                xx = x.clone().requires_grad_()
                saved = []
                with torch.autograd.graph.saved_tensors_hooks(lambda t: (saved.append(t.clone()), t)[1], lambda t: t):
                    y = op(xx * 1.0, *args)
                name = y.grad_fn.name()
                assert ('FewbitPackedBackward' in name) == direct, (name, direct)
                y.backward(gy, retain_graph=True)
                g1 = xx.grad.clone()
                y.backward(gy)                                              # second pass: the graph was retained once
                assert torch.equal(xx.grad, 2 * g1)
                with pytest.raises(RuntimeError, match='second time'):
                    y.backward(gy)
                got.append((y.detach().clone(), [t for t in saved if t.dtype == torch.uint8][0], g1))
            with pytest.raises(RuntimeError, match='leaf Variable'):
                torch.ops.fewbit.silu(x.clone().requires_grad_(), inner, levels)
            results[direct] = got
    finally:
        fewbit_amd.autograd_route('direct_node', prev)
    for a, b in zip(results[True], results[False]):
        assert all(torch.equal(u, v) for u, v in zip(a, b))


def test_in_place_on_a_whole_view_on_the_host_takes_the_base_route_and_stays_correct():
    """the host operators share the view handling of the GPU ones: in place on a view that covers its whole base (the 3-D
    output of nn.Linear) modifies the base, returns a fresh view of it, and every alias keeps working"""
    import fewbit_amd
    torch.manual_seed(0)
    lin = torch.nn.Linear(16, 24)
    x = torch.randn(4, 5, 16)
    inner, levels = store.get_inner('silu', 3, torch.device('cpu'), torch.float32)
    wgt = torch.randn(4, 5, 24)

    def run(shape3d):
        lin.zero_grad(set_to_none=True)
        h = lin(x if shape3d else x.view(-1, 16))
        assert h._is_view() == shape3d
        out = torch.ops.fewbit.silu(h, inner, levels)
        assert out.data_ptr() == h.data_ptr() and torch.equal(out.detach(), h.detach())
        ((out * (wgt if shape3d else wgt.view(-1, 24))).sum() + 0.5 * h.sum()).backward()
        return out.detach().reshape(-1, 24).clone(), lin.weight.grad.clone(), out.grad_fn.name()

    y3, g3, name3 = run(True)
    y2, g2, _ = run(False)
    assert torch.equal(y3, y2) and torch.allclose(g3, g2, rtol=1e-5, atol=1e-6)
    if fewbit_amd.autograd_internals():
        assert 'ViewBackward' in name3 or 'Reshape' in name3 or 'View' in name3, name3      # the fresh view, not CopySlices
        prev = fewbit_amd.autograd_route('base_dirty', False)
        try:
            y3g, g3g, _ = run(True)                       # autograd's general in-place-on-view route: the same numbers
        finally:
            fewbit_amd.autograd_route('base_dirty', prev)
        assert torch.equal(y3g, y2) and torch.allclose(g3g, g2, rtol=1e-5, atol=1e-6)
