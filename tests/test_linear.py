"""Randomized linear layers, DCT and variance utilities -- the reference's own checks restated
(fewbit/modules/linear_test.py, fewbit/fft_test.py): forward equals nn.Linear, the input and bias gradients are exact,
the weight gradient is an unbiased estimate (mean over repeats within 10 % of the exact one)."""
import itertools

import numpy as np
import pytest
import scipy.fft
import torch

import fewbit
from fewbit.fft import dct, idct
from fewbit.functional import catch_gradients, GradientStorage, linear_crs, linear_grp, linear_randomized
from fewbit.modules.linear import LinearCRS, LinearGRP, RandomizedLinear


def reference_linear(module):
    clone = torch.nn.Linear(module.in_features, module.out_features, module.bias is not None)
    clone.load_state_dict({k: v.clone() for k, v in module.state_dict().items()})
    return clone


def mean_grads(module, xs, repeat):
    acc_w = torch.zeros_like(module.weight)
    acc_b = torch.zeros_like(module.bias) if module.bias is not None else None
    for _ in range(repeat):
        module.zero_grad()
        xs.grad = None
        ys = module(xs)
        ys.backward(torch.ones_like(ys))
        acc_w += module.weight.grad
        if acc_b is not None:
            acc_b += module.bias.grad
    return xs.grad, acc_w / repeat, None if acc_b is None else acc_b / repeat


@pytest.mark.parametrize('ctor', (LinearCRS, LinearGRP))
@pytest.mark.parametrize('bias', (False, True))
def test_forward_equals_linear(ctor, bias):
    torch.manual_seed(42)
    module = ctor(8, 4, bias, proj_dim=64)
    xs = torch.randn(128, 8)
    with torch.no_grad():
        rel = torch.linalg.norm(module(xs) - reference_linear(module)(xs)) / torch.linalg.norm(module(xs))
    assert rel.item() <= 1e-6


@pytest.mark.parametrize('ctor,kwargs', [(LinearCRS, {}), (LinearGRP, {}), (LinearGRP, {'matmul': 'rademacher'}),
                                         (LinearGRP, {'matmul': 'dct'}), (LinearGRP, {'matmul': 'dft'})])
@pytest.mark.parametrize('bias', (False, True))
def test_backward_is_unbiased(ctor, kwargs, bias):
    torch.manual_seed(42)
    module = ctor(in_features=64, out_features=32, bias=bias, proj_dim=32, **kwargs)
    exact = reference_linear(module)
    xs = torch.randn(128, 64, requires_grad=True)
    gi, gw, gb = mean_grads(module, xs, 1500)
    ri, rw, rb = mean_grads(exact, xs, 1)
    assert (torch.linalg.norm(gi - ri) / torch.linalg.norm(ri)).item() <= 1e-6
    assert (torch.linalg.norm(gw - rw) / torch.linalg.norm(rw)).item() <= 1e-1
    if bias:
        assert (torch.linalg.norm(gb - rb) / torch.linalg.norm(rb)).item() <= 1e-6


def test_saved_tensor_is_the_projection_and_generator_replays():
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(7)
    m = RandomizedLinear(16, 8, proj_dim_ratio=0.25, generator=gen)
    assert RandomizedLinear is LinearGRP and linear_randomized is linear_grp
    x = torch.randn(4, 10, 16, requires_grad=True)            # 40 rows -> 10 projected rows
    y = m(x)
    saved = [t for t in y.grad_fn.saved_tensors if t.shape[-1] == 16 and t.shape[0] != 8]
    assert saved and saved[0].shape == (10, 16)
    y.sum().backward()
    g1 = m.weight.grad.clone()
    # same generator state -> same sketch -> same estimate
    m.zero_grad()
    gen.manual_seed(7)
    m(x).sum().backward()
    assert torch.equal(g1, m.weight.grad)
    assert 'matmul=gaussian' in repr(m) and 'nopairs=' in repr(LinearCRS(4, 4))
    with pytest.raises(ValueError):
        linear_grp(x, m.weight, m.bias)                                   # neither proj_dim nor ratio
    with pytest.raises(ValueError):
        linear_grp(x, m.weight, m.bias, 0.5, None, None, None, 'haar')    # unknown projection
    with pytest.raises(ValueError):
        linear_grp(x, m.weight, m.bias, 0.5, None, 2, 4)                  # max < min
    assert linear_crs(x, m.weight, m.bias, 5).shape == (4, 10, 8)


def test_dct_matches_scipy():
    x = np.random.default_rng(0).standard_normal((3, 37, 4))
    for kind, norm, dim, n in itertools.product((2, 3), ('backward', 'forward', 'ortho'), (0, 1, -1), (None, 20, 50)):
        assert np.abs(dct(torch.tensor(x), kind, n, dim, norm).numpy() - scipy.fft.dct(x, kind, n, dim, norm)).max() < 1e-10
        assert np.abs(idct(torch.tensor(x), kind, n, dim, norm).numpy() - scipy.fft.idct(x, kind, n, dim, norm)).max() < 1e-10
    x32 = torch.tensor(x, dtype=torch.float32)
    assert torch.allclose(idct(dct(x32, norm='ortho', dim=1), norm='ortho', dim=1), x32, atol=1e-5)
    with pytest.raises(ValueError):
        dct(x32, norm='unitary')


def test_gradient_catcher_and_variance_estimator():
    torch.manual_seed(1)
    store = GradientStorage()
    x = torch.randn(6, 3, requires_grad=True)
    store.forward(x)
    (catch_gradients(x * 2.0, store) * torch.arange(3.0)).sum().backward()
    assert torch.equal(store.input, x.detach()) and torch.equal(store.grad_output, torch.arange(3.0).expand(6, 3))
    seen = []
    est = fewbit.variance.VarianceEstimator(LinearGRP(8, 4, proj_dim=5), lambda *a: seen.append(a))
    inp = torch.randn(20, 8)
    est(inp).square().sum().backward()
    corr, var_sgd, var_rmm = est.variance
    assert len(seen) == 1 and seen[0][3] == 0 and 0.0 <= corr.item() <= 1.0 and var_rmm.item() >= 0.0
    g = est.state.grad_output
    want = (torch.linalg.norm(inp)**2 * torch.linalg.norm(g)**2 - torch.linalg.norm(inp.T @ g)**2) / 5
    assert torch.allclose(var_rmm, want, rtol=1e-5)


def test_sketch_dtype_keeps_the_estimate_unbiased():
    """sketch_dtype=bfloat16 (the MI355X choice for fp32 layers): same forward, exact input gradient, and the mean of
    the weight-gradient estimates still converges to the exact gradient."""
    torch.manual_seed(3)
    module = LinearGRP(64, 32, proj_dim=32, sketch_dtype=torch.bfloat16)
    exact = reference_linear(module)
    xs = torch.randn(128, 64, requires_grad=True)
    gi, gw, gb = mean_grads(module, xs, 1500)
    ri, rw, rb = mean_grads(exact, xs, 1)
    assert (torch.linalg.norm(gi - ri) / torch.linalg.norm(ri)).item() <= 1e-6
    assert (torch.linalg.norm(gw - rw) / torch.linalg.norm(rw)).item() <= 1e-1
    assert gw.dtype == torch.float32


def test_replay_survives_a_dtype_change_between_forward_and_backward():
    """CPU autocast: fp32 input in forward, bf16 grad_output in backward.  The Gaussian sketch must be replayed from
    the forward's draw dtype (randn's stream depends on the dtype) -- otherwise the estimate is pure noise."""
    torch.manual_seed(5)
    module = LinearGRP(48, 24, proj_dim=24)
    xs = torch.randn(96, 48)
    exact = (torch.ones(96, 24).T @ xs)                        # d/dW of sum(linear(x)) is ones^T x
    acc = torch.zeros_like(exact)
    runs = 400
    for _ in range(runs):
        module.zero_grad()
        with torch.autocast('cpu', dtype=torch.bfloat16):
            module(xs).sum().backward()
        acc += module.weight.grad
    rel = (torch.linalg.norm(acc / runs - exact) / torch.linalg.norm(exact)).item()
    assert rel <= 0.15, rel                                     # was ~1.0 (uncorrelated S in forward and backward)


# ---- fixtures generated by importing the python reference (tests/golden/gen_linear_golden.py) -----------------------------
@pytest.fixture(scope='module')
def lref():
    import numpy as np
    from helpers import GOLDEN
    with np.load(GOLDEN / 'linear_ref.npz') as z:
        return {k: z[k].copy() for k in z.files}


def test_dct_and_variance_formulas_equal_the_reference(lref):
    for i in range(5):
        x = torch.from_numpy(lref[f'dct{i}_x'])
        dim = int(lref[f'dct{i}_dim'])
        # (the reference builds its twiddle factors in complex64 whatever the input dtype -- fewbit/fft.py:28-29,66-67 --,
        # so it is only single-precision accurate; its idct with norm='backward' additionally scales the DC term by
        # 1/sqrt(2) twice (fft.py:55-56,67-68) and is off from scipy.fft.idct by a constant x0-dependent offset: a reference
        # defect, not reproduced -- there this implementation is pinned to scipy instead)
        for norm in ('backward', 'ortho'):
            assert torch.allclose(fewbit.fft.dct(x, dim=dim, norm=norm), torch.from_numpy(lref[f'dct{i}_{norm}']), rtol=1e-6, atol=1e-6)
        assert torch.allclose(fewbit.fft.idct(x, dim=dim, norm='ortho'), torch.from_numpy(lref[f'idct{i}_ortho']), rtol=1e-6, atol=1e-6)
        want = torch.from_numpy(scipy.fft.idct(x.numpy(), axis=dim, norm='backward'))
        assert torch.allclose(fewbit.fft.idct(x, dim=dim, norm='backward'), want, rtol=1e-12, atol=1e-12)
        off = torch.from_numpy(lref[f'idct{i}_backward']) - want
        assert float(off.abs().max()) > 1e-5 and float((off - off.mean(dim=dim, keepdim=True)).abs().max()) < 1e-6
    a, b = torch.from_numpy(lref['var_input']), torch.from_numpy(lref['var_output'])
    V = fewbit.variance
    for got, key in ((V.estimate_correlation(a, b), 'var_correlation'), (V.estimate_variance_sgd(a, b), 'var_sgd'),
                     (V.estimate_variance_sgd(a, b, 12), 'var_sgd_bs12'), (V.estimate_variance_rmm(a, b), 'var_rmm'),
                     (V.estimate_variance_rmm(a, b, 5), 'var_rmm_bs5')):
        assert torch.allclose(got, torch.from_numpy(lref[key]), rtol=1e-12), key


def _draw_stats(make, x, w, bias, gy, exact, draws):
    acc = torch.zeros_like(exact, dtype=torch.float64)
    msd, first = 0.0, None
    for _ in range(draws):
        xi, wi, bi = x.clone().requires_grad_(), w.clone().requires_grad_(), bias.clone().requires_grad_()
        y = make(xi, wi, bi)
        y.backward(gy)
        if first is None:
            first = (y.detach(), xi.grad, bi.grad)
        acc += wi.grad.double()
        msd += float(((wi.grad - exact)**2).sum())
    return first, (acc / draws).float(), msd / draws


def test_randomized_linear_against_the_reference_run(lref):
    """Deterministic outputs (y, grad_input, grad_bias) equal the reference's; the random weight-gradient estimators have
    the reference's DISTRIBUTION where the reference works: same mean (the exact gradient) and the same mean squared
    deviation for the Gaussian and Rademacher sketches and for column-row sampling.  Reference defects not reproduced:
    its 'dct' estimator is scaled by proj_dim^2 (fewbit/functional/linear.py:121-123 multiplies by the factor it should
    divide by) and its 'dft' estimator raises on current torch (real @ complex, :215); here both are unbiased."""
    x, w, bias, gy = (torch.from_numpy(lref[k]) for k in ('lin_x', 'lin_w', 'lin_b', 'lin_gy'))
    exact = torch.from_numpy(lref['lin_exact_gw'])
    p, draws = int(lref['lin_proj_dim']), 1500
    torch.manual_seed(11)
    assert list(lref['grp_reference_raises']) == ['dft']
    for kind in ('gaussian', 'rademacher', 'dct', 'dft'):
        first, mean, msd = _draw_stats(lambda xi, wi, bi: fewbit.functional.linear_grp(xi, wi, bi, proj_dim=p, matmul=kind),
                                       x, w, bias, gy, exact, draws)
        rel = float(torch.linalg.norm(mean - exact) / torch.linalg.norm(exact))
        assert rel <= 0.08, (kind, rel)                                         # unbiased (reference: 0.026 over 4000 draws)
        if kind != 'dft':
            for got, key in zip(first, ('y', 'gx', 'gb')):
                assert torch.allclose(got, torch.from_numpy(lref[f'grp_{kind}_{key}']), rtol=1e-5, atol=1e-5), (kind, key)
        if kind in ('gaussian', 'rademacher'):
            ref_mean = torch.from_numpy(lref[f'grp_{kind}_mean_gw'])
            assert float(torch.linalg.norm(ref_mean - exact) / torch.linalg.norm(exact)) <= 0.05
            assert abs(msd / float(lref[f'grp_{kind}_msd']) - 1.0) <= 0.10, (kind, msd, float(lref[f'grp_{kind}_msd']))
        if kind == 'dct':                                                       # the reference's mean is p^2 x the exact gradient
            ref_mean = torch.from_numpy(lref['grp_dct_mean_gw'])
            scale = float((ref_mean * exact).sum() / (exact * exact).sum())
            assert abs(scale / p**2 - 1.0) <= 0.05, scale
    nopairs = int(lref['crs_nopairs'])
    first, mean, msd = _draw_stats(lambda xi, wi, bi: fewbit.functional.linear_crs(xi, wi, bi, nopairs), x, w, bias, gy, exact, draws)
    assert float(torch.linalg.norm(mean - exact) / torch.linalg.norm(exact)) <= 0.08
    for got, key in zip(first, ('y', 'gx', 'gb')):
        assert torch.allclose(got, torch.from_numpy(lref[f'crs_{key}']), rtol=1e-5, atol=1e-5), key
    assert abs(msd / float(lref['crs_msd']) - 1.0) <= 0.10


# ---- single draws with the reference's own sketch matrix injected (tests/golden/gen_linear_golden.py draw) ----------------
def check_injected_draws(device, rtol=2e-4):
    """Shared by the CPU test below and the -m gpu test (tests/test_gpu_linear.py): with the matrix the reference drew,
    forward, input gradient, bias gradient AND the weight gradient equal the reference's up to fp32 GEMM rounding
    (rtol on the largest entry: the products are sums of <= 256 fp32 terms in another order)."""
    import numpy as np
    from helpers import GOLDEN
    from fewbit_amd.linear import inject_sketch
    with np.load(GOLDEN / 'linear_draw_ref.npz') as z:
        ref = {k: torch.from_numpy(z[k].copy()) for k in z.files}
    checked = 0
    for name in ('small', 'mid', 'ragged'):
        p = int(ref[f'{name}_p'])
        for kind in ('gaussian', 'rademacher'):
            S = ref[f'{name}_{kind}_S']
            if kind == 'rademacher':
                S = 2 * S - 1                               # the reference keeps {0,1} - 0.5 and folds the 4 into the scale
            x, w, b = (ref[f'{name}_{k}'].clone().to(device).requires_grad_() for k in ('x', 'w', 'b'))
            with inject_sketch(S.to(device)):
                y = linear_grp(x, w, b, proj_dim=p, matmul=kind)
                y.backward(ref[f'{name}_gy'].to(device))
            for got, key in ((y.detach(), 'y'), (x.grad, 'gx'), (b.grad, 'gb'), (w.grad, 'gw')):
                want = ref[f'{name}_{kind}_{key}']
                err = float((got.cpu() - want).abs().max() / want.abs().max())
                assert got.device.type == torch.device(device).type and err <= rtol, (name, kind, key, err)
                checked += 1
    return checked


def test_single_draws_with_the_reference_sketch_injected():
    assert check_injected_draws('cpu', rtol=1e-5) == 24
