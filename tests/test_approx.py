"""Table pipeline: the generator must reproduce the tables the path runs on.
Expected values: the reference's own unit test (fewbit/approx_test.py:22-69) and the built-in tables themselves
(produced by the reference with tools/quantize-builtins.sh: seed 42, -M 100000, eps 1e-6)."""
import numpy as np
import pytest
import scipy.special
import torch
import torch.nn.functional as F

from fewbit_amd.approx import StepWiseFunction, approximate, estimate_error
from fewbit_amd.cli import main as cli_main, quantize
from fewbit_amd.store import BUILTIN_TABLES, StepwiseStore, store


def gelu(x):
    return 0.5 * x * (1 + scipy.special.erf(x / np.sqrt(2)))


def gelu_grad(x):
    return 0.5 * (1 + scipy.special.erf(x / np.sqrt(2))) + x * np.exp(-0.5 * x**2) / np.sqrt(2 * np.pi)


# expected optimum from fewbit/approx_test.py:24-34
BORDERS = np.array([-2.39798704e+00, -7.11248159e-01, -3.26290283e-01, -1.55338428e-04, 3.26182064e-01, 7.10855860e-01,
                    2.39811567e+00])
LEVELS = np.array([-0.00260009, -0.08883533, 0.1251944, 0.37204148, 0.6277958, 0.87466175, 1.08880716, 1.00259936])
KW = dict(fn=gelu_grad, fn_prim=gelu, cardinality=8, parity=False, max_iters=2000, beps=1e-6, leps=1e-6, domain=(-100, 100),
          random_state=42)


def test_approximate_like_reference():
    fn, info = approximate(**KW)
    assert info['status'] == 'converged'
    assert np.linalg.norm(fn.borders[1:-1] - BORDERS) < 0.05          # assertAlmostEqual(places=1)
    assert np.linalg.norm(fn.levels - LEVELS) < 0.005                 # places=2


def test_approximate_parity_like_reference():
    fn, info = approximate(**{**KW, 'cardinality': 4, 'parity': True, 'domain': (0, 100)})
    assert info['status'] == 'converged'
    assert np.linalg.norm(fn.borders[:-1] - BORDERS[3:]) < 0.05
    assert np.linalg.norm(fn.levels - LEVELS[4:]) < 0.005
    with pytest.raises(ValueError):
        approximate(**{**KW, 'parity': True})


@pytest.mark.parametrize('spec,bits', [('torch.nn.functional:gelu', 1), ('torch.nn.functional:gelu', 2),
                                       ('torch.nn.functional:gelu', 3), ('torch.nn.functional:gelu', 4),
                                       ('torch.nn.functional:silu', 2), ('torch.nn.functional:silu', 4),
                                       ('torch:tanh', 3), ('torch:sigmoid', 3), ('torch.nn.functional:softsign', 3)])
def test_generator_reproduces_builtin_tables(spec, bits):
    quant = quantize(bits, spec, seed=42, max_iters=100000, border_error=1e-6, level_error=1e-6)
    name = spec.split(':')[1]
    with np.load(BUILTIN_TABLES) as z:
        assert np.abs(quant.borders - z[f'{name}{bits:02d}-borders']).max() < 1e-9
        assert np.abs(quant.levels - z[f'{name}{bits:02d}-levels']).max() < 1e-9


def test_stepwise_function_semantics():
    b, l = store.get('gelu', 3)
    f = StepWiseFunction(b.numpy(), l.numpy())
    xs = np.linspace(-5, 5, 1001)
    assert np.array_equal(f(xs), l.numpy()[np.searchsorted(b.numpy()[1:-1], xs, side='left')])
    assert f(np.array([b[3].item()]))[0] == l[2].item()               # a point on a border takes the lower level
    assert f.card == 8 and np.allclose(np.cumsum(f.steps), l.numpy())
    assert 'nosteps=8' in repr(f) and str(f).count('\n') == 8
    with pytest.raises(ValueError):
        StepWiseFunction(np.zeros(3), np.zeros(3))


@pytest.mark.parametrize('name', ('celu', 'elu', 'gelu', 'hardswish', 'logsigmoid', 'mish', 'selu', 'sigmoid', 'silu',
                                  'softplus', 'softsign', 'tanh', 'tanhshrink'))
def test_builtin_tables_approximate_the_derivative(name):
    # the bar of the reference's TestContinousFunctions (fewbit/functional/activations_test.py:91-109): L2 error of
    # the 3-bit table against the autograd derivative <= 1e-1
    fn = getattr(torch, name) if name in ('sigmoid', 'tanh') else getattr(F, name)

    def deriv(xs):
        t = torch.tensor(xs, requires_grad=True)
        fn(t).backward(torch.ones_like(t))
        return t.grad.numpy()

    b, l = store.get(name, 3)
    err, per_piece = estimate_error(deriv, StepWiseFunction(b.numpy(), l.numpy()), 1e-2)
    assert 0 <= err <= 1e-1 and per_piece.shape == (8,)


def test_cli_writes_the_store_format(tmp_path, capsys):
    out = tmp_path / 'tables.npz'
    cli_main(['--log-level', 'error', 'quantize', '-s', '42', '-M', '100000', '-o', str(out), '2', 'torch.nn.functional:gelu'])
    cli_main(['--log-level', 'error', 'quantize', '-s', '42', '-M', '100000', '-o', str(out), '1', 'torch:tanh'])   # update
    with np.load(out) as z:
        assert sorted(z.files) == ['gelu02-borders', 'gelu02-levels', 'tanh01-borders', 'tanh01-levels']
    s = StepwiseStore().load(out)
    assert len(s) == 2
    assert torch.equal(s.get('gelu', 2)[0], store.get('gelu', 2)[0].double().float())
    cli_main(['version'])
    assert 'fewbit version' in capsys.readouterr().out
    with pytest.raises(SystemExit):
        cli_main(['--log-level', 'error', 'quantize', '-M', '3', '-s', '1', '3', 'torch.nn.functional:gelu'])    # cannot converge
