"""The drop-in surface on the GPU: torch.ops.fewbit.*, fewbit.functional.*, fewbit.<Module> -- the reference's own
GPU tests (fewbit/functional/activations_test.py) restated, plus in-place/alias/stream/memory behaviour."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import fewbit
import oracle
from helpers import DTYPES, GOLDEN, assert_bit_equal, forward_value_ok, from_raw

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_native_loaded():
    import fewbit_amd
    assert fewbit_amd.native_loaded(), fewbit_amd.native_error()


@pytest.mark.parametrize('name,args,kwargs', [
    ('hardshrink', (), {}), ('hardshrink', (), {'lambd': 1.0}), ('hardsigmoid', (), {}), ('hardtanh', (), {}),
    ('hardtanh', (), {'min_val': -2.0, 'max_val': 2.0}), ('leaky_relu', (), {}), ('leaky_relu', (), {'negative_slope': 0.5}),
    ('relu', (), {}), ('relu6', (), {}), ('softshrink', (), {}), ('softshrink', (), {'lambd': 1.0}),
    ('threshold', (1.0, 3.0), {})])
def test_stepwise_functions_like_reference(name, args, kwargs):
    # fewbit/functional/activations_test.py:16-68
    xs = torch.linspace(-5, 5, 101).to(DEV)
    gs = torch.ones_like(xs)
    ps = xs.clone().requires_grad_()
    ys = getattr(F, name)(ps, *args, **kwargs)
    ys.backward(gs)
    qs = xs.clone().requires_grad_()
    zs = getattr(fewbit.functional, name)(qs.clone(), *args, **kwargs)
    zs.backward(gs)
    assert torch.linalg.norm(zs - ys).item() < 1e-6
    assert torch.linalg.norm(ps.grad - qs.grad).item() < 1e-6


@pytest.mark.parametrize('name', fewbit.functional.CONTINOUS)
def test_continuous_functions_like_reference(name):
    # fewbit/functional/activations_test.py:78-148: forward tracks torch on the device; here the returned gradient
    # is additionally pinned to the table (the reference only estimated the table's L2 error)
    ref = getattr(torch, name) if name in ('sigmoid', 'tanh') else getattr(F, name)
    xs = torch.linspace(-5, 5, 101).to(DEV)
    ys = ref(xs)
    qs = xs.clone().requires_grad_()
    zs = getattr(fewbit.functional, name)(qs.clone())
    assert torch.linalg.norm(zs - ys).item() <= 1e-6
    zs.backward(torch.ones_like(xs))
    borders, levels = fewbit.functional.store.get(name, 3, DEV, torch.float32)
    assert torch.equal(qs.grad, levels[torch.bucketize(xs, borders[1:-1].contiguous())])


@pytest.mark.parametrize('dt', list(DTYPES))
@pytest.mark.parametrize('bits', (1, 2, 3, 4))
def test_gelu_module_matches_oracle(dt, bits):
    dtype = DTYPES[dt]
    g = torch.Generator().manual_seed(bits)
    x = (torch.randn(3, 1000, 7, generator=g) * 1.5).to(dtype)
    gy = torch.randn(3, 1000, 7, generator=g).to(dtype)
    borders, levels = fewbit.functional.store.get('gelu', bits, 'cpu', dtype)
    y_o, s_o, _ = oracle.quantize('gelu', x.flatten(), borders[1:-1])
    gx_o = oracle.quantize_backward(gy.flatten(), s_o, levels).view_as(x)
    xd = x.to(DEV).requires_grad_()
    inp = xd.clone()
    out = fewbit.GELU(bits=bits)(inp)
    assert out.data_ptr() == inp.data_ptr()                 # in place, returns the alias (Tensor(a!))
    out.backward(gy.to(DEV))
    assert_bit_equal(xd.grad.cpu(), gx_o, f'gelu {dt} bits={bits} grad')
    assert forward_value_ok(x, out.detach().cpu(), y_o.view_as(x)).all()


def test_saved_for_backward_is_the_packed_state():
    n = 1 << 16
    x = torch.randn(n, device=DEV, requires_grad=True)
    with fewbit.memory_usage_hooks() as vanilla:
        F.gelu(x).sum().backward()
    with fewbit.memory_usage_hooks() as ours:
        fewbit.functional.gelu(x.clone(), bits=3).sum().backward()
    assert vanilla.forward == 4 * n
    assert ours.forward == 3 * n // 8 + 8 * 4               # packed codes + the level table
    with fewbit.memory_usage_hooks() as one:
        fewbit.functional.relu(x.clone()).sum().backward()
    assert one.forward == n // 8


def test_raw_quantize_ops_against_golden():
    with np.load(GOLDEN / 'quantize_ref.npz') as z:
        for dt, dtype in DTYPES.items():
            key = f'gelu03_{dt}_1001'
            x, gy = from_raw(z[key + '_x'], dtype), from_raw(z[key + '_gy'], dtype)
            b, l = from_raw(z[f'gelu03_{dt}_borders'], dtype), from_raw(z[f'gelu03_{dt}_levels'], dtype)
            xd = x.to(DEV)
            y, state = torch.ops.fewbit.quantize(xd, b.to(DEV))
            assert y.data_ptr() != xd.data_ptr() and torch.equal(xd.cpu().view(torch.int16 if dt != 'f32' else torch.int32),
                                                                 x.view(torch.int16 if dt != 'f32' else torch.int32))
            want = torch.from_numpy(z[key + '_state'])
            assert_bit_equal(state.cpu()[:want.numel()], want, key)
            gx = torch.ops.fewbit.quantize_backward(gy.to(DEV), state, l.to(DEV))
            assert_bit_equal(gx.cpu(), from_raw(z[key + '_gx'], dtype), key)


def test_errors_and_contracts():
    x = torch.randn(64, 64, device=DEV)
    b, l = fewbit.functional.store.get('gelu', 3, DEV, torch.float32)
    with pytest.raises(RuntimeError, match='contiguous'):
        torch.ops.fewbit.gelu(x.t(), b[1:-1], l)
    with pytest.raises(RuntimeError, match='dtype'):
        torch.ops.fewbit.gelu(x.clone(), b[1:-1].half(), l)
    with pytest.raises(RuntimeError, match='lesser'):
        torch.ops.fewbit.gelu(x.clone(), b[1:-2], l)
    with pytest.raises(RuntimeError):
        torch.ops.fewbit.gelu(x.double(), b[1:-1].double(), l.double())
    host = torch.randn(8)
    assert torch.equal(torch.ops.fewbit.gelu(host.clone(), b[1:-1].cpu(), l.cpu()), F.gelu(host))   # CPU key (tests/test_host_ops.py)
    with pytest.raises(RuntimeError):
        torch.ops.fewbit.gelu(x.clone(), b[1:-1].cpu(), l.cpu())          # tables and input on different devices
    with pytest.raises(RuntimeError):
        torch.ops.fewbit.gelu(host.clone(), b[1:-1], l)
    with pytest.raises(RuntimeError, match='parity'):                     # a shift without a parity means nothing
        torch.ops.fewbit.stepwise(x.clone(), b[1:-1], l, None, [1, 0])
    leaf = torch.randn(8, device=DEV, requires_grad=True)
    with pytest.raises(RuntimeError):                                     # in-place on a leaf, as in the reference
        fewbit.functional.gelu(leaf)


def test_custom_table_module_and_stream():
    b, l = fewbit.functional.store.get('tanh', 4, DEV, torch.float32)
    m = fewbit.Stepwise(b, l).to(DEV)
    x = torch.randn(4099, device=DEV)
    xr = x.clone().requires_grad_()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = m(xr.clone())
        out.backward(torch.ones_like(out))
    s.synchronize()
    assert torch.equal(out.detach(), x)
    assert torch.equal(xr.grad, l[torch.bucketize(x, b[1:-1].contiguous())])
    # custom borders/values through the functional keyword interface
    xr2 = x.clone().requires_grad_()
    fewbit.functional.gelu(xr2.clone(), borders=b, values=l).sum().backward()
    assert torch.equal(xr2.grad, xr.grad)


def test_map_module_training_step_and_memory():
    torch.manual_seed(0)
    def mlp():
        return torch.nn.Sequential(torch.nn.Linear(256, 1024), torch.nn.GELU(), torch.nn.Linear(1024, 256)).to(DEV)
    base, ours = mlp(), mlp()
    ours.load_state_dict(base.state_dict())
    fewbit.map_module(ours, lambda m, p: fewbit.GELU(bits=3) if isinstance(m, torch.nn.GELU) else m)
    x = torch.randn(4096, 256, device=DEV)
    with fewbit.memory_usage_hooks() as mb:
        base(x).square().mean().backward()
    with fewbit.memory_usage_hooks() as mo:
        ours(x).square().mean().backward()
    saved = 4096 * 1024 * 4 - 4096 * 1024 * 3 // 8
    assert mb.forward - mo.forward >= saved - 4096
    g0, g1 = base[0].weight.grad, ours[0].weight.grad
    cos = torch.nn.functional.cosine_similarity(g0.flatten(), g1.flatten(), dim=0).item()
    assert cos > 0.97                                        # 3-bit derivative: close, not equal
    assert torch.allclose(base[2].weight.grad, ours[2].weight.grad, rtol=1e-4, atol=1e-6)   # downstream of the activation


def test_view_inputs_run_out_of_place_and_match():
    """nn.Linear on a 3-D input returns a VIEW; an in-place op on it would make autograd rebase the view (CopySlices:
    a zero-fill and four full-size copies per backward).  The functional layer then writes a fresh tensor instead;
    values and gradients must be the same as on the in-place route."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 256).to(DEV)
    x = torch.randn(8, 16, 64, device=DEV)
    h = lin(x)
    assert h._is_view()
    out = fewbit.GELU(bits=3)(h)
    assert out.data_ptr() != h.data_ptr() and not out._is_view()
    out.sum().backward()
    g_view = lin.weight.grad.clone()
    lin.weight.grad = None
    h2 = lin(x.view(-1, 64))                       # 2-D input: plain tensor, in-place route
    assert not h2._is_view()
    out2 = fewbit.GELU(bits=3)(h2)
    assert out2.data_ptr() == h2.data_ptr()
    out2.sum().backward()
    assert torch.equal(out.view(-1, 256), out2) and torch.allclose(g_view, lin.weight.grad, rtol=1e-5, atol=1e-6)
    r = torch.randn(4, 32, 8, device=DEV, requires_grad=True)
    v = (r * 1.0).view(4, 256)
    y = fewbit.functional.relu(v)
    assert y.data_ptr() != v.data_ptr() and torch.equal(y, F.relu(v))
    y.sum().backward()
    assert torch.equal(r.grad.view(4, 256), (v > 0).float())


def _graph_nodes(fn):
    seen, stack, names = set(), [fn], []
    while stack:
        f = stack.pop()
        if f is None or f in seen:
            continue
        seen.add(f)
        names.append(type(f).__name__)
        stack += [n for n, _ in f.next_functions]
    return names


ROUTES = {'default': {}, 'function_node': {'direct_node': False}, 'returns_self': {'fresh_view': False},
          'general_view_route': {'base_dirty': False}, 'all_public_api': {'direct_node': False, 'base_dirty': False}}


@pytest.fixture
def route(request):
    """switch the operator library's autograd routes for one test (fewbit_amd.autograd_route), restore afterwards"""
    import fewbit_amd
    if request.param != 'default' and not fewbit_amd.autograd_internals():
        pytest.skip('operator library built without the internal-API routes: only the public-API route exists')
    # every switch is set explicitly ('default' = all on), so that a FEWBIT_NO_* variable in the environment of the test run
    # does not change what a parameter means
    wanted = {**{'direct_node': True, 'base_dirty': True, 'fresh_view': True}, **ROUTES[request.param]} if fewbit_amd.autograd_internals() else {}
    prev = {k: fewbit_amd.autograd_route(k, v) for k, v in wanted.items()}
    yield request.param
    for k, v in prev.items():
        fewbit_amd.autograd_route(k, v)


@pytest.mark.parametrize('route', list(ROUTES), indirect=True)
def test_raw_operator_in_place_on_a_linear_output_view(route):
    """The reference's own caller route (benchmark/bench-roberta.py:138-147): the RAW operator, in place, on the 3-D output
    of nn.Linear -- a view of its 2-D addmm result.  The operator modifies the view's BASE (torch_ops.cpp, whole_view_base),
    so autograd builds no CopySlices node; values, gradients and the saved bytes equal the 2-D route's.  Every autograd
    route of the library (hand-written node / torch::autograd::Function, base write on or off, fresh view or `self`
    returned) must give the same bytes and gradients; with the base write off autograd's general machinery (CopySlices) runs."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 256).to(DEV)
    x = torch.randn(8, 16, 64, device=DEV)
    wgt = torch.randn(8, 16, 256, device=DEV)
    inner, levels = fewbit.functional.store.get_inner('gelu', 3, torch.device(DEV), torch.float32)
    for op_name, args in (('gelu', (inner, levels)), ('relu', ()), ('leaky_relu', (0.1,))):
        op = getattr(torch.ops.fewbit, op_name)
        lin.zero_grad(set_to_none=True)
        h = lin(x)
        assert h._is_view()
        want_ptr = h.data_ptr()
        with fewbit.memory_usage_hooks() as usage:
            out = op(h, *args)
            assert out.data_ptr() == want_ptr and out.shape == h.shape           # in place: the same memory comes back
            # (one gradient contribution through `out`, one through the OLD python object `h`, which stays usable: a sum of
            # two terms is the same in either order, so the routes can be compared bit for bit)
            loss = (out * wgt).sum() + 0.5 * h.sum()
        nodes = _graph_nodes(out.grad_fn)
        general = ROUTES[route].get('base_dirty') is False
        assert ('CopySlices' in nodes) == general, (route, nodes)
        if route == 'default':
            assert 'AsStridedBackward0' not in nodes and 'ViewBackward0' in nodes, nodes       # the returned view: a reshape
        direct = ROUTES[route].get('direct_node', True)
        if not general:                                   # (CopySlices hides the operator's node inside itself)
            assert any('FewbitPackedBackward' in n for n in _graph_names(out.grad_fn)) == direct
        loss.backward()
        g_view, gb_view = lin.weight.grad.clone(), lin.bias.grad.clone()
        saved_view = usage.forward
        # the 2-D route: plain tensor, plain in-place
        lin.zero_grad(set_to_none=True)
        h2 = lin(x.view(-1, 64))
        assert not h2._is_view()
        with fewbit.memory_usage_hooks() as usage2:
            out2 = op(h2, *args)
            loss2 = (out2 * wgt.view(-1, 256)).sum() + 0.5 * h2.sum()
        loss2.backward()
        assert torch.equal(out.view(-1, 256), out2), op_name
        assert torch.equal(g_view, lin.weight.grad) and torch.equal(gb_view, lin.bias.grad), (route, op_name)
        if not general:
            assert saved_view == usage2.forward, (op_name, saved_view, usage2.forward)


def _graph_names(fn):
    seen, stack, names = set(), [fn], []
    while stack:
        f = stack.pop()
        if f is None or f in seen:
            continue
        seen.add(f)
        names.append(f.name())
        stack += [n for n, _ in f.next_functions]
    return names


@pytest.mark.parametrize('route', ['default', 'function_node'], indirect=True)
def test_both_node_routes_save_the_same_bytes_and_give_the_same_gradients(route):
    """every operator family x in place / out of place x 3 dtypes: packed state and gradient vs the oracle under both
    autograd routes, plus autograd's own errors (leaf in place, second backward)"""
    # (non-leaf inputs are made with .clone(), not `* 1.0`: ATen's own fp16 multiply turns -0 into +0 in its remainder loop
    # -- scratch/dbg_sign.py -- and MulBackward0 would put that into the gradient the oracle is compared with)
    import oracle
    g = torch.Generator().manual_seed(3)
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        x = (torch.randn(4099, generator=g) * 2).to(dtype)
        gy = torch.randn(4099, generator=g).to(dtype)
        inner, levels = fewbit.functional.store.get_inner('gelu', 3, torch.device('cpu'), dtype)
        _, state_o, _ = oracle.quantize('gelu', x, inner)
        gx_o = oracle.quantize_backward(gy, state_o, levels)
        _, bits_o = oracle.stepwise1_forward('relu', x)
        gr_o = oracle.stepwise1_backward('relu', gy, bits_o)
        for inplace in (True, False):
            xd = x.to(DEV).requires_grad_()
            seen = []
            with torch.autograd.graph.saved_tensors_hooks(lambda t: (seen.append(t), t)[1], lambda t: t):
                if inplace:
                    y = torch.ops.fewbit.gelu(xd.clone(), inner.to(DEV), levels.to(DEV))
                else:
                    y = torch.ops.fewbit.continuous_out(xd.clone(), inner.to(DEV), levels.to(DEV), 2, 0.0, 0.0)
            assert ('FewbitPackedBackward' in y.grad_fn.name()) == (route == 'default')
            assert torch.equal([t for t in seen if t.dtype == torch.uint8][0].cpu(), state_o)
            y.backward(gy.to(DEV), retain_graph=True)
            assert_bit_equal(xd.grad, gx_o, f'gelu {dtype} inplace={inplace} {route}')
            y.backward(gy.to(DEV))
            with pytest.raises(RuntimeError, match='second time'):
                y.backward(gy.to(DEV))
            xr = x.to(DEV).requires_grad_()
            yr = torch.ops.fewbit.relu(xr.clone()) if inplace else torch.ops.fewbit.stepwise1_out(xr.clone(), 4, 0.0, 0.0)
            yr.backward(gy.to(DEV))
            assert_bit_equal(xr.grad, gr_o, f'relu {dtype} inplace={inplace} {route}')
        with pytest.raises(RuntimeError, match='leaf Variable'):
            torch.ops.fewbit.gelu(x.to(DEV).requires_grad_(), inner.to(DEV), levels.to(DEV))


def test_in_place_on_a_view_keeps_in_place_semantics_for_later_users():
    """After `op(v)` (in place) every alias of the memory -- the view `v`, its base, another view of the base -- holds the
    activation values AND carries the operator in its history, exactly as with any in-place op."""
    inner, levels = fewbit.functional.store.get_inner('silu', 4, torch.device(DEV), torch.float32)
    r = torch.randn(6, 40, device=DEV, requires_grad=True)

    def run(route):
        r.grad = None
        base = r * 1.0                                   # non-leaf 2-D tensor
        v = base.view(6, 5, 8)
        if route == 'view':
            out = torch.ops.fewbit.silu(v, inner, levels)
            assert out.data_ptr() == base.data_ptr()
        else:                                            # reference point: in place on the base itself
            torch.ops.fewbit.silu(base, inner, levels)
        # users of the aliases AFTER the op: the old view, the base, a fresh view
        total = (v * 2.0).sum() + (base * 3.0).sum() + (base.view(-1)[::2] * 5.0).sum()
        total.backward()
        return base.detach().clone(), r.grad.clone()

    y_view, g_view = run('view')
    y_base, g_base = run('base')
    assert torch.equal(y_view, y_base) and torch.equal(g_view, g_base)
    assert torch.allclose(y_view, F.silu(r.detach()), rtol=1e-5, atol=1e-6)
    code = torch.searchsorted(inner, r.detach().flatten()).view(6, 40)
    w = torch.full((6, 40), 5.0, device=DEV)
    w.view(-1)[1::2] = 0.0
    assert torch.allclose(g_view, levels[code] * (w + 5.0))


def test_views_autograd_refuses_to_modify_keep_raising():
    """A whole-tensor view that comes out of a multi-output view op (unbind of a size-1 dimension) covers its base too, but
    autograd forbids modifying such views in place; the operator must not route around that check."""
    inner, levels = fewbit.functional.store.get_inner('gelu', 3, torch.device(DEV), torch.float32)
    r = torch.randn(1, 4096, device=DEV, requires_grad=True)
    v = (r * 1.0).unbind(0)[0]
    assert v._is_view() and v.numel() == r.numel()
    with pytest.raises(RuntimeError, match='view'):
        torch.ops.fewbit.gelu(v, inner, levels)
    with pytest.raises(RuntimeError, match='view'):
        torch.ops.fewbit.relu((r * 1.0).unbind(0)[0])


def test_partial_views_take_the_general_route_and_stay_correct():
    """A view that does not cover its whole base (a slice) cannot use the base shortcut: autograd's own in-place-on-view
    machinery (CopySlices) runs, and the result must still be right."""
    inner, levels = fewbit.functional.store.get_inner('gelu', 2, torch.device(DEV), torch.float32)
    r = torch.randn(8, 512, device=DEV, requires_grad=True)
    base = r * 1.0
    v = base[2:6]                                        # contiguous slice: rows 2..5
    assert v._is_view() and v.is_contiguous()
    out = torch.ops.fewbit.gelu(v, inner, levels)
    assert out.data_ptr() == v.data_ptr()
    assert 'CopySlices' in _graph_nodes(base.grad_fn)
    (base * 1.0).sum().backward()
    code = torch.searchsorted(inner, r.detach()[2:6].flatten()).view(4, 512)
    want = torch.ones(8, 512, device=DEV)
    want[2:6] = levels[code]
    assert torch.equal(r.grad, want)
    assert torch.equal(base.detach()[:2], r.detach()[:2]) and torch.equal(base.detach()[6:], r.detach()[6:])


def test_hip_graph_capture_and_replay():
    """The launches never synchronise or touch the host, so a whole forward+backward (operator level, autograd
    included) can be captured into a hipGraph once and replayed on new data: the replay must give the oracle's
    packed bytes and gradients for the data that is in the static buffers at replay time."""
    from fewbit_amd import cabi
    n, k = 40_000 + 3, 3
    dtype = torch.bfloat16
    b, l = fewbit.functional.store.get('gelu', k, DEV, dtype)
    inner = b[1:-1].contiguous()
    xs = torch.zeros(n, device=DEV, dtype=dtype)             # static graph inputs
    gs = torch.zeros(n, device=DEV, dtype=dtype)
    y = torch.empty_like(xs)
    gx = torch.empty_like(xs)
    st = torch.empty(cabi.state_nbytes(n, k), dtype=torch.uint8, device=DEV)
    xa = torch.zeros(n, device=DEV, dtype=dtype, requires_grad=True)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                            # warm-up outside the capture (allocator, lazy init)
        for _ in range(2):
            cabi.quantize_forward('gelu', xs, inner, out=y, state=st)
            cabi.quantize_backward(gs, st, l, out=gx)
            out = fewbit.functional.gelu(xa * 1.0, bits=k)
            ga, = torch.autograd.grad(out, xa, gs)
    torch.cuda.current_stream().wait_stream(side)

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cabi.quantize_forward('gelu', xs, inner, out=y, state=st)       # C-ABI launches on the capturing stream
        cabi.quantize_backward(gs, st, l, out=gx)
        out = fewbit.functional.gelu(xa * 1.0, bits=k)                  # operator + autograd node
        ga, = torch.autograd.grad(out, xa, gs)

    for seed in (1, 2):
        g = torch.Generator().manual_seed(seed)
        xh = (torch.randn(n, generator=g) * 2).to(dtype)
        gh = torch.randn(n, generator=g).to(dtype)
        xs.copy_(xh)
        gs.copy_(gh)
        with torch.no_grad():
            xa.copy_(xh)
        graph.replay()
        torch.cuda.synchronize()
        y_o, st_o, _ = oracle.quantize('gelu', xh, inner.cpu())
        gx_o = oracle.quantize_backward(gh, st_o, l.cpu())
        assert torch.equal(st.cpu(), st_o)
        assert_bit_equal(gx.cpu(), gx_o)
        assert_bit_equal(ga.cpu(), gx_o)
        assert torch.equal(out.detach().cpu().view(torch.int16), y.cpu().view(torch.int16))
        assert forward_value_ok(xh, y.cpu(), y_o).all()


@pytest.mark.parametrize('dt', list(DTYPES))
def test_stepwise_parity_and_shift_on_gpu(dt):
    """fewbit.Stepwise(borders, levels, parity, shift) on the GPU against the oracle: even -> folded key, 2 bits for 4
    half-line levels; odd -> the mirrored plain table (8 levels, 3 bits) built in fp32 and rounded once."""
    dtype = DTYPES[dt]
    b = torch.tensor([0.5, 1.0, 2.0]).to(dtype)
    l = torch.tensor([1.0, 0.6, 0.3, 0.1]).to(dtype)
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(5000, generator=g) * 2,
                   torch.tensor([0.0, -0.0, 0.5, -0.5, 1.5, 2.5, -1.0, float('inf'), -float('inf')])]).to(dtype)
    gy = torch.randn(x.numel(), generator=g).to(dtype)
    for sx, sy in ((0.0, 0.5), (1.0, 0.25)):
        # even
        m = fewbit.Stepwise(b, l, parity=True, shift=(sx, sy)).to(DEV)
        xr = x.to(DEV).requires_grad_()
        y = m(xr.clone())          # (torch's own fp16 `x * 1.0` turns -0 into +0 in its tail)
        y.backward(gy.to(DEV))
        _, st, _ = oracle.quantize('identity_fold', x, b, sx)
        assert_bit_equal(xr.grad.cpu(), oracle.quantize_backward(gy, st, l), f'even {dt} {sx}')
        assert_bit_equal(y.detach().cpu(), x, f'even y {dt}')
        # odd
        full_b = torch.cat([sx - b.float().flip(0), torch.tensor([sx]), b.float() + sx]).to(dtype)
        full_l = torch.cat([2.0 * sy - l.float().flip(0), l.float()]).to(dtype)
        xr = x.to(DEV).requires_grad_()
        y = fewbit.functional.stepwise(xr.clone(), b.to(DEV), l.to(DEV), parity=False, shift=(sx, sy))
        y.backward(gy.to(DEV))
        _, st, k = oracle.quantize('identity', x, full_b)
        assert k == 3
        assert_bit_equal(xr.grad.cpu(), oracle.quantize_backward(gy, st, full_l), f'odd {dt} {sx}')
        # the reference's own schema (integer shift)
        xr2 = x.to(DEV).requires_grad_()
        y2 = torch.ops.fewbit.stepwise(xr2.clone(), b.to(DEV), l.to(DEV), True, [int(sx), 0])
        y2.backward(gy.to(DEV))
        _, st, _ = oracle.quantize('identity_fold', x, b, float(int(sx)))
        assert_bit_equal(xr2.grad.cpu(), oracle.quantize_backward(gy, st, l), f'schema {dt}')
    # view input -> out-of-place operator
    base = torch.randn(4, 64, device=DEV, dtype=dtype, requires_grad=True)
    v = (base * 1.0).view(256)
    out = fewbit.functional.stepwise(v, b.to(DEV), l.to(DEV), parity=True)
    assert out.data_ptr() != v.data_ptr()
    out.sum().backward()
    assert base.grad.shape == base.shape


@pytest.mark.parametrize('matmul', ('gaussian', 'rademacher', 'dct', 'dft'))
def test_randomized_linear_on_gpu(matmul):
    """RandomizedLinear on the GPU in bf16: forward == nn.Linear, exact input gradient, weight-gradient estimate
    correlated with the exact one, and only the projected rows are kept for backward (with a little activation stack
    on top: fewbit.GELU saves 3 bits per element)."""
    torch.manual_seed(0)
    dtype = torch.bfloat16
    lin = fewbit.RandomizedLinear(256, 512, proj_dim_ratio=0.25, matmul=matmul, device=DEV, dtype=dtype)
    ref = torch.nn.Linear(256, 512, device=DEV, dtype=dtype)
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(2048, 256, device=DEV, dtype=dtype, requires_grad=True)
    y = lin(x)
    assert torch.equal(y, ref(x))
    kept = [t for t in y.grad_fn.saved_tensors if t.dim() == 2 and t.shape == (512, 256) and t.data_ptr() != lin.weight.data_ptr()]
    assert kept, [t.shape for t in y.grad_fn.saved_tensors]
    g = torch.randn_like(y)
    acc = torch.zeros_like(lin.weight, dtype=torch.float32)
    for _ in range(64):
        lin.zero_grad()
        x.grad = None
        lin(x).backward(g)
        acc += lin.weight.grad.float()
    gi = x.grad.clone()
    x.grad = None
    ref(x).backward(g)
    assert torch.equal(gi, x.grad)
    exact = ref.weight.grad.float()
    rel = (torch.linalg.norm(acc / 64 - exact) / torch.linalg.norm(exact)).item()
    assert rel < 0.35, rel                                   # one draw: ~sqrt(rows/p) = 2; mean of 64: ~0.25
    out = fewbit.GELU(bits=3)(lin(x))
    out.sum().backward()


def test_torch_free_host_program_on_the_c_abi():
    """examples/cabi_demo.cpp: plain C++ + the HIP runtime + libfewbit_hip.so (no torch, no python) -- forward and
    backward of one million elements, one random-projection product and one sampled cosine transform, checked on the host by the program itself."""
    import subprocess
    from helpers import ROOT
    exe = ROOT / 'examples' / 'cabi_demo'
    assert exe.exists(), 'build it with `make -C fewbit_amd/csrc demo`'
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'mismatches: codes 0, gradients 0, forward 0' in out.stdout
    assert '(Rademacher, seed 123456789abcdef): mismatches 0' in out.stdout      # the random-projection kernel, S rebuilt on the host
    assert 'seed in device memory (counter 6 -> 7): differences from the seed by value 0' in out.stdout
    assert 'sampled DCT of 256 x 6, 5 rows picked (workspace 67632 bytes): mismatches 0' in out.stdout                 # vs the cosine sum in double
    assert 'seeded call against the explicit one: mismatches 0' in out.stdout and out.stdout.rstrip().endswith('mismatches 0')


def test_inference_mode_and_no_grad_run_the_kernels():
    """Under torch.inference_mode() the Autograd keys are excluded: the plain CUDA-key registration must serve the
    call (the reference, registered for AutogradCUDA only, raises there)."""
    from fewbit_amd.store import store
    x = torch.randn(5000, device=DEV, dtype=torch.bfloat16)
    inner, levels = store.get_inner('gelu', 3, torch.device(DEV), torch.bfloat16)
    want = fewbit.functional.gelu(x.clone().requires_grad_().clone(), bits=3).detach()
    with torch.inference_mode():
        y = torch.ops.fewbit.gelu(x.clone(), inner, levels)
        assert torch.equal(y.view(torch.int16), want.view(torch.int16))
        z = fewbit.GELU(bits=3)(x.clone())
        assert torch.equal(z.view(torch.int16), want.view(torch.int16))
        r = torch.ops.fewbit.relu(x.clone())
        assert torch.equal(r, F.relu(x))
        o = torch.ops.fewbit.continuous_out(x, inner, levels, 2)          # FEWBIT_GELU
        assert torch.equal(o.view(torch.int16), want.view(torch.int16)) and o.data_ptr() != x.data_ptr()
    with torch.no_grad():
        y = fewbit.functional.silu(x.clone(), bits=2)
    assert not y.requires_grad and forward_value_ok(x.cpu(), y.cpu(), F.silu(x.float()).to(torch.bfloat16).cpu()).all()


def test_strided_inputs_are_gathered_and_left_intact():
    """chunk(2, -1) of a GEGLU, a transpose, channels_last: non-contiguous views go through the out-of-place operators
    after a gather (F.gelu accepts them; the flat-memory kernels need the copy)."""
    base = torch.randn(64, 96, device=DEV, dtype=torch.float16)
    for view in (base.chunk(2, -1)[1], base.t(), base[:, ::2]):
        assert not view.is_contiguous()
        keep = view.clone()
        v = view.detach().requires_grad_()
        y = fewbit.functional.gelu(v, bits=3)
        assert y.shape == view.shape and torch.equal(view, keep)             # input untouched
        ok = forward_value_ok(keep.cpu(), y.detach().cpu(), F.gelu(keep.float()).half().cpu())
        assert ok.all()
        y.sum().backward()
        borders, levels = fewbit.functional.store.get('gelu', 3, 'cpu', torch.float16)
        code = torch.searchsorted(borders[1:-1].float(), keep.cpu().float())
        assert torch.equal(v.grad.cpu(), levels[code])
        r = fewbit.functional.relu(view.detach().requires_grad_())
        assert torch.equal(r, F.relu(keep)) and torch.equal(view, keep)
    x4 = torch.randn(2, 8, 5, 5, device=DEV).to(memory_format=torch.channels_last)
    assert torch.allclose(fewbit.functional.silu(x4, bits=4), F.silu(x4), atol=1e-6)


def test_state_moves_between_host_and_device_operators():
    """The host operators (AutogradCPU/CPU keys) pack the same bytes as the kernels: a state produced on one side is
    consumed on the other."""
    from fewbit_amd.store import store
    g = torch.Generator().manual_seed(9)
    for dtype in (torch.float32, torch.bfloat16):
        x = (torch.randn(10007, generator=g) * 2).to(dtype)
        gy = torch.randn(10007, generator=g).to(dtype)
        inner, levels = store.get_inner('gelu', 3, torch.device('cpu'), dtype)
        _, st_host = torch.ops.fewbit.quantize(x, inner)
        _, st_dev = torch.ops.fewbit.quantize(x.to(DEV), inner.to(DEV))
        assert torch.equal(st_host, st_dev.cpu())
        gx_dev_from_host = torch.ops.fewbit.quantize_backward(gy.to(DEV), st_host.to(DEV), levels.to(DEV))
        gx_host_from_dev = torch.ops.fewbit.quantize_backward(gy, st_dev.cpu(), levels)
        assert_bit_equal(gx_dev_from_host.cpu(), gx_host_from_dev)


def test_inplace_write_without_a_node_still_bumps_the_version():
    """A tensor that does not require grad takes the node-free path; it is still overwritten in place, so a graph that
    saved its old value (for another operand's gradient) must fail loudly in backward."""
    x = torch.randn(256, device=DEV)
    w = torch.randn(256, device=DEV, requires_grad=True)
    y = x * w                                           # saves x for dw
    v0 = x._version
    out = fewbit.functional.relu(x)
    assert out.data_ptr() == x.data_ptr() and x._version > v0 and not out.requires_grad
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        y.sum().backward()
    x2 = torch.randn(256, device=DEV)
    v0 = x2._version
    fewbit.functional.gelu(x2, bits=2)
    assert x2._version > v0


def test_c_abi_validation_mode():
    """FEWBIT_HIP_VALIDATE=1 (read once per process, hence the subprocess): host pointers and under-sized buffers are
    refused with an error code and a message instead of faulting on the GPU; valid calls still run."""
    import subprocess
    import sys
    import textwrap
    from helpers import ROOT
    code = textwrap.dedent('''
        import ctypes, sys, torch
        sys.path.insert(0, %r)
        from fewbit_amd import cabi
        L = cabi.lib()
        x = torch.randn(4096, device='cuda'); y = torch.empty_like(x)
        b = torch.tensor([-1.0, 0.0, 1.0], device='cuda')
        st = torch.empty(cabi.state_nbytes(4096, 2), dtype=torch.uint8, device='cuda')
        s = torch.cuda.current_stream().cuda_stream
        call = lambda xp, yp, sp, bp: L.fewbit_hip_quantize_forward(2, 0, xp, yp, sp, 4096, bp, 3, 0.0, 0.0, s)
        assert call(x.data_ptr(), y.data_ptr(), st.data_ptr(), b.data_ptr()) == 0
        host = torch.randn(4096)
        assert call(host.data_ptr(), y.data_ptr(), st.data_ptr(), b.data_ptr()) == -1
        assert b'x' in L.fewbit_hip_last_error() and b'device' in L.fewbit_hip_last_error()
        # extents are those of the underlying hipMalloc (a caching allocator's sub-blocks share one): allocate directly
        hip = ctypes.CDLL('libamdhip64.so')
        small = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(small), ctypes.c_size_t(100)) == 0
        assert call(x.data_ptr(), y.data_ptr(), small.value, b.data_ptr()) == -1
        assert b'state needs 1024 bytes' in L.fewbit_hip_last_error(), L.fewbit_hip_last_error()
        hip.hipFree(small)
        lv = torch.tensor([0.0, 0.3, 0.7, 1.0], device='cuda')
        assert L.fewbit_hip_quantize_backward(0, x.data_ptr(), st.data_ptr(), y.data_ptr(), 4096, lv.data_ptr(), 4, s) == 0
        big = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(big), ctypes.c_size_t(4096 * 4)) == 0
        assert L.fewbit_hip_quantize_backward(0, big.value, st.data_ptr(), y.data_ptr(), 4096, lv.data_ptr(), 4, s) == 0
        assert L.fewbit_hip_quantize_backward(0, big.value, st.data_ptr(), y.data_ptr(), 8192, lv.data_ptr(), 4, s) == -1
        assert b'gy needs' in L.fewbit_hip_last_error()
        torch.cuda.synchronize()
        print('validated')
    ''' % str(ROOT))
    import os
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, FEWBIT_HIP_VALIDATE='1'), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and 'validated' in r.stdout, r.stdout + r.stderr


def test_concurrent_streams_and_threads_one_process_many_devices():
    """SURVEY 8(e), first form: ONE process driving every device of the box with per-device streams.  Host thread i works on
    device i % device_count() (all of them on the one device of a 1-GPU box) on its own stream: the activation path (per-device
    launch-geometry and occupancy caches, first use included when this test runs alone), a random-projection product and a sampled
    cosine transform of 32768 rows, whose kernels need the per-device opt-in to more than 64 KiB of LDS.  The library keeps no per-call global state, so every thread
    must get the bytes of the serial run on the first device, whichever device it ran on."""
    import threading
    from fewbit_amd import cabi
    from fewbit_amd.store import store
    ndev = torch.cuda.device_count()
    devices = [torch.device('cuda', i % ndev) for i in range(4)]
    first = devices[0]
    sizes = (8 * 1024 * 1024 + 13, 1_000_003, 4096, 6 * 1024 * 1024 + 512)
    host = []
    for i, n in enumerate(sizes):
        dtype = (torch.bfloat16, torch.float32, torch.float16, torch.bfloat16)[i]
        g = torch.Generator().manual_seed(100 + i)
        host.append(((torch.randn(n, generator=g) * 1.5).to(dtype), torch.randn(n, generator=g).to(dtype), dtype))
    m_host = torch.randn(2048, 1024, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16)
    d_host = torch.randn(32768, 40, generator=torch.Generator().manual_seed(10))
    d_idx = torch.randint(0, 32768, (300, ), generator=torch.Generator().manual_seed(11))

    def tables(dtype, dev):
        b, l = store.get('gelu', 3, dev, dtype)
        return b[1:-1].contiguous(), l

    def sketch_on(dev, stream=None):
        # Gaussian, 1024 features, S generated in the product kernel: the 128 x 512 tile, 160 KiB of LDS (hipFuncSetAttribute per device)
        plan = cabi.describe_sketch('gaussian', 2048, 1024, 300, torch.bfloat16, device=dev)
        assert plan['lds_bytes'] > 65536 and plan['s_fragment_bytes'] == 0, plan
        return cabi.sketch('gaussian', m_host.to(dev), 300, 77, 1.0 / 300, stream=stream)

    serial = []
    cabi.tune(sketch_materialise=0)
    try:
        for x, gy, dtype in host:
            b, l = tables(dtype, first)
            y, st = cabi.quantize_forward('gelu', x.to(first), b)
            serial.append((y.cpu(), st.cpu(), cabi.quantize_backward(gy.to(first), st, l).cpu()))
        serial_sketch = sketch_on(first).cpu()
        serial_dct = cabi.sampled_dct(d_host.to(first), d_idx.to(first)).cpu()
        torch.cuda.synchronize(first)
        out, errors, ran_on = [None] * len(host), [], [None] * len(host)

        def run(i):
            try:
                dev = devices[i]
                s = torch.cuda.Stream(device=dev)
                x, gy, dtype = host[i]
                x, gy = x.to(dev), gy.to(dev)
                b, l = tables(dtype, dev)
                with torch.cuda.stream(s):
                    for _ in range(20):
                        y, st = cabi.quantize_forward('gelu', x, b, stream=s.cuda_stream)
                        gx = cabi.quantize_backward(gy, st, l, stream=s.cuda_stream)
                    p = sketch_on(dev, stream=s.cuda_stream)
                    dc = cabi.sampled_dct(d_host.to(dev), d_idx.to(dev), stream=s.cuda_stream)
                s.synchronize()
                assert y.device == dev and p.device == dev and dc.device == dev
                # the plan the library reports for THIS device is the one it used there (geometry cached per device index)
                assert cabi.describe_forward('gelu', dtype, x.numel(), 7, device=dev)['blocks'] > 0
                ran_on[i] = dev.index
                out[i] = (y.cpu(), st.cpu(), gx.cpu(), p.cpu(), dc.cpu())
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=run, args=(i,)) for i in range(len(host))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        cabi.tune(sketch_materialise=-1)
    assert not errors, errors
    assert ran_on == [d.index for d in devices] and (ndev == 1 or any(ran_on)), ran_on      # a non-zero device index whenever one exists
    for (y0, s0, g0), (y1, s1, g1, p1, dc1) in zip(serial, out):
        assert torch.equal(s0, s1)
        assert torch.equal(y0.view(torch.uint8), y1.view(torch.uint8)) and torch.equal(g0.view(torch.uint8), g1.view(torch.uint8))
        assert torch.equal(serial_sketch.view(torch.uint8), p1.view(torch.uint8))
        assert torch.equal(serial_dct.view(torch.uint8), dc1.view(torch.uint8))


def test_autocast_and_activation_checkpointing():
    """The module inside common training wrappers: CUDA autocast (bf16 activations into an fp32-parameter model) and
    torch.utils.checkpoint (the in-place forward is re-run during backward) give the gradients of the plain run."""
    from torch.utils.checkpoint import checkpoint
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(256, 1024), fewbit.GELU(bits=3), torch.nn.Linear(1024, 64)).to(DEV)
    x = torch.randn(512, 256, device=DEV)

    def grads(fn):
        net.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = fn(x)
        out.float().square().mean().backward()
        return [p.grad.clone() for p in net.parameters()]

    plain = grads(net)
    ckpt = grads(lambda t: checkpoint(net, t, use_reentrant=False))
    for a, b in zip(plain, ckpt):
        assert torch.equal(a, b)
    # and the activation really went through the bf16 kernels with a packed state
    seen = []
    with torch.autograd.graph.saved_tensors_hooks(lambda t: (seen.append((t.dtype, t.numel())), t)[1], lambda t: t):
        with torch.autocast('cuda', dtype=torch.bfloat16):
            net(x)
    assert (torch.uint8, 3 * 512 * 1024 // 8) in seen
    # reference gradient: same net with the exact GELU derivative replaced by the table -> close to vanilla
    van = torch.nn.Sequential(net[0], torch.nn.GELU(), net[2])
    net.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        van(x).float().square().mean().backward()
    gv = net[0].weight.grad
    cos = torch.nn.functional.cosine_similarity(gv.flatten(), plain[0].flatten(), dim=0)
    assert cos > 0.97, cos


def test_module_fast_path_follows_the_store_and_the_input_layout():
    """fewbit.<Module>.forward binds its table casts per (device, dtype): a table replaced through store.add() must be
    seen by the next call, and views / strided inputs must take the out-of-place route, as through fewbit.functional."""
    from fewbit_amd.store import store
    name, bits = 'tanh', 2
    old = store.get(name, bits)
    m = fewbit.Tanh(bits=bits)
    x = torch.linspace(-3, 3, 4096, device=DEV)
    try:
        a = x.clone().requires_grad_()
        m(a.clone()).sum().backward()
        borders = torch.tensor([-100.0, -1.0, 0.0, 1.0, 100.0], dtype=torch.float64)
        levels = torch.tensor([0.125, 0.25, 0.5, 1.0], dtype=torch.float64)
        store.add(name, bits, (borders, levels))
        b = x.clone().requires_grad_()
        m(b.clone()).sum().backward()
        want = levels.float().to(DEV)[torch.bucketize(x, borders[1:-1].float().to(DEV))]
        assert torch.equal(b.grad, want) and not torch.equal(a.grad, b.grad)
        # a view (row slice of a bigger buffer) and a strided tensor: input untouched, fresh output, same gradient
        base = torch.randn(8, 4096, device=DEV, requires_grad=True)
        for select in (lambda t: t[3], lambda t: t.t()[:, 3]):
            base.grad = None
            hidden = base * 1.0                                      # non-leaf; the selections below are views of it
            inp = select(hidden)
            keep = inp.detach().clone()
            out = m(inp)
            assert torch.equal(inp.detach(), keep) and out.data_ptr() != inp.data_ptr()
            out.sum().backward()
            assert torch.equal(base.grad[3], levels.float().to(DEV)[torch.bucketize(keep, borders[1:-1].float().to(DEV))])
            assert int((base.grad != 0).sum()) == int((base.grad[3] != 0).sum())
    finally:
        store.add(name, bits, old)
