"""Host-side logic and the drop-in surface, no GPU: API parity with the reference (fixture captured by importing
the reference, tests/golden/api_surface.json), table store, error behaviour, host-tensor path, helpers, and that the
C-ABI library loads and exports every symbol include/fewbit_hip.h declares."""
import ctypes
import inspect
import json
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import fewbit
import fewbit_amd
import oracle
from fewbit_amd import cabi
from helpers import DTYPES, GOLDEN, ROOT, assert_bit_equal


@pytest.fixture(scope='module')
def api():
    return json.loads((GOLDEN / 'api_surface.json').read_text())


def test_native_libraries_load_and_export_the_abi():
    header = (ROOT / 'include' / 'fewbit_hip.h').read_text()
    declared = sorted(set(re.findall(r'\b(fewbit_hip_\w+)\s*\(', header)))
    assert declared == sorted(cabi.SYMBOLS)
    lib = ctypes.CDLL(str(cabi.LIB_PATH))
    for sym in declared:
        assert hasattr(lib, sym), sym
    # pure host helpers of the ABI may be called without a GPU
    L = cabi.lib()
    header_version = int(re.search(r"#define FEWBIT_HIP_ABI_VERSION (\d+)", header).group(1))
    assert L.fewbit_hip_abi_version() == header_version == cabi.ABI_VERSION == 5
    assert [L.fewbit_hip_bitwidth(v) for v in (2, 3, 4, 5, 8, 9, 16, 256)] == [1, 2, 2, 3, 3, 4, 4, 8]
    assert L.fewbit_hip_state_nbytes(16777216, 3) == 6291456 and L.fewbit_hip_state_nbytes(9, 3) == 6
    assert fewbit_amd.native_loaded(), fewbit_amd.native_error()
    for name in fewbit.functional.STEPWISE + fewbit.functional.CONTINOUS + ('quantize', 'quantize_backward'):
        assert hasattr(torch.ops.fewbit, name), name


# the frozen C-ABI (FEWBIT_HIP_ABI_VERSION 5): what `nm -D` must list for libfewbit_hip.so -- exactly this, nothing else
FROZEN_ABI = '''
fewbit_hip_abi_version fewbit_hip_last_error fewbit_hip_bitwidth fewbit_hip_state_nbytes
fewbit_hip_quantize_forward fewbit_hip_quantize_backward fewbit_hip_stepwise1_forward fewbit_hip_stepwise1_backward
fewbit_hip_pack_codes fewbit_hip_unpack_codes
fewbit_hip_describe_quantize_forward fewbit_hip_describe_quantize_backward fewbit_hip_describe_stepwise1_forward
fewbit_hip_describe_stepwise1_backward fewbit_hip_tune
fewbit_hip_sketch_workspace fewbit_hip_sketch fewbit_hip_sketch_device_seed fewbit_hip_sketch_next_seed fewbit_hip_sketch_mix_seed
fewbit_hip_sketch_matrix fewbit_hip_sketch_describe fewbit_hip_philox4x32 fewbit_hip_xoshiro128pp
fewbit_hip_sampled_dct_workspace fewbit_hip_sampled_dct fewbit_hip_sampled_dct_seeded fewbit_hip_sampled_rows
'''.split()


def test_exported_symbol_list_is_the_frozen_abi():
    """`nm -D --defined-only libfewbit_hip.so` lists the frozen interface and nothing else: no measurement hook, no kernel stub, no
    template instantiation (fewbit_amd/csrc/fewbit_hip.map).  A new entry point means a new FEWBIT_HIP_ABI_VERSION and a new list here."""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', str(cabi.LIB_PATH)], check=True, capture_output=True, text=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == sorted(FROZEN_ABI), sorted(set(exported) ^ set(FROZEN_ABI))
    assert sorted(cabi.SYMBOLS) == sorted(FROZEN_ABI)
    header = (ROOT / 'include' / 'fewbit_hip.h').read_text()
    assert 'sketch_tune_' not in header.split('#define FEWBIT_HIP_ABI_VERSION')[1]      # (the version history may name them)


def test_tune_is_the_single_hook_and_validates_its_keys():
    """host-only: the six random-projection settings are keys of fewbit_hip_tune; unknown keys and bad values are refused by name"""
    for key in ('sketch_slices', 'sketch_waves', 'sketch_halves', 'sketch_convert', 'sketch_partials', 'sketch_materialise'):
        cabi.tune(**{key: -1})
    for key, bad in (('sketch_waves', 5), ('sketch_halves', 3), ('sketch_slices', 0), ('sketch_convert', 2), ('sketch_partials', 3),
                     ('sketch_materialise', 2), ('no_such_key', 1), ('u_lut', 1), ('lut_block', 512)):
        with pytest.raises(cabi.FewbitHipError, match=key):
            cabi.tune(**{key: bad})
    # "whenever possible" keeps the caps: a 4 GiB S is not written to memory even when asked for (rows 2^21, proj 2^10 would need ~4 GiB)
    try:
        ws = cabi.lib().fewbit_hip_sketch_workspace
        shapes = ((1 << 21, 1024, 1 << 10), (16384, 256, 3276))      # a ~4 GiB S; one column tile (nothing to share: stays fused)
        cabi.tune(sketch_materialise=1)
        asked = [ws(1, 2, *shape) for shape in shapes] + [ws(1, 2, 16384, 3072, 3276)]
        cabi.tune(sketch_materialise=0)
        never = [ws(1, 2, *shape) for shape in shapes] + [ws(1, 2, 16384, 3072, 3276)]
        assert asked[:2] == never[:2] and asked[0] < (1 << 30)
        assert asked[2] - never[2] >= 3276 * 16384 * 2                # (where the path applies the fragments are in the workspace)
    finally:
        cabi.tune(sketch_materialise=-1)


def test_sampled_dct_workspace_is_a_host_side_formula():
    """no GPU needed: ceil(features / 64) * rows * 256 + 2048 + 8 * proj (rounded up to 16) bytes for 2^k rows in [256, 262144] or 3 x 2^k rows in
    [768, 49152], 0 = no kernel for this shape"""
    assert cabi.sampled_dct_workspace_bytes(16384, 768, 3276) == 12 * 16384 * 256 + 2048 + 8 * 3276
    assert cabi.sampled_dct_workspace_bytes(65536, 70, 1, torch.float32) == 2 * 65536 * 256 + 2048 + 16
    assert cabi.sampled_dct_workspace_bytes(12288, 64, 5) == 12288 * 256 + 2048 + 48
    assert cabi.sampled_dct_workspace_bytes(262144, 64, 8) == 262144 * 256 + 2048 + 64
    assert cabi.sampled_dct_workspace_bytes(20480, 64, 8) == 20480 * 256 + 2048 + 64
    for rows in (0, 48, 128, 255, 3000, 1792, 81920, 98304, 524288):
        assert cabi.sampled_dct_workspace_bytes(rows, 64, 10) == 0
    assert cabi.sampled_dct_workspace_bytes(1024, 0, 10) == 0 == cabi.sampled_dct_workspace_bytes(1024, 64, 0)
    assert cabi.sampled_dct_workspace_bytes(1024, 64, 10, torch.float64) == 0


def test_every_file_under_profiles_is_indexed():
    """profiles/README.md is a generated table (file -> one line -> the DESIGN / EXPERIMENTS section that quotes it); a file without a row fails"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(ROOT / 'tools' / 'profiles_index.py'), '--check'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_design_document_stays_a_design():
    """at most 400 lines of at most 160 characters (the experiment log lives in EXPERIMENTS.md)"""
    lines = (ROOT / 'DESIGN.md').read_text().split('\n')
    assert len(lines) <= 400, len(lines)
    assert not [i + 1 for i, l in enumerate(lines) if len(l.encode()) > 160]


def test_operator_schemas_match_reference():
    # fewbit/fewbit.cc:10-37
    want = {
        'gelu': 'fewbit::gelu(Tensor(a!) self, Tensor bounds, Tensor levels) -> Tensor(a!)',
        'relu': 'fewbit::relu(Tensor(a!) self) -> Tensor(a!)',
        'leaky_relu': 'fewbit::leaky_relu(Tensor(a!) self, float negative_slope=0.01) -> Tensor(a!)',
        'hardtanh': 'fewbit::hardtanh(Tensor(a!) self, float min_val=-1., float max_val=1.) -> Tensor(a!)',
        'threshold': 'fewbit::threshold(Tensor(a!) self, float threshold, float value) -> Tensor(a!)',
        'softplus': 'fewbit::softplus(Tensor(a!) self, Tensor bounds, Tensor levels, float beta=1., float threshold=20.) -> Tensor(a!)',
        'stepwise': 'fewbit::stepwise(Tensor(a!) self, Tensor bounds, Tensor levels, bool? parity=None, int[2]? shift=None) -> Tensor(a!)',
    }
    for name, schema in want.items():
        assert str(getattr(torch.ops.fewbit, name).default._schema) == schema


def test_module_surface_matches_reference(api):
    for name, entry in api['modules'].items():
        cls = getattr(fewbit, name)
        if name == 'Stepwise':
            assert [p for p in inspect.signature(cls.__init__).parameters] == ['self', 'borders', 'levels', 'parity', 'shift']
            continue
        assert str(inspect.signature(cls.__init__)) == entry['init'], name
        made = cls(1.0, 3.0, bits=3) if name == 'Threshold' else cls(bits=3)
        assert repr(made) == entry['repr'], name
        if 'repr_default' in entry:
            assert repr(cls()) == entry['repr_default']
    assert sorted(fewbit.modules.__all__) == sorted(api['modules'])


def test_functional_surface_matches_reference(api):
    for name, ref_sig in api['functional'].items():
        fn = getattr(fewbit.functional, name)
        params = inspect.signature(fn).parameters
        assert list(params)[0] == 'input'
        if name in fewbit.functional.CONTINOUS:
            # same keyword-only interface; positional part is the torch function's (the reference could not
            # introspect some of them, e.g. softplus, and silently lost their parameters -- SURVEY 2.2 defect 11)
            kw = [(p.name, p.default) for p in params.values() if p.kind == p.KEYWORD_ONLY]
            assert kw == [('bits', None), ('borders', None), ('values', None)], name
            if ref_sig and '/' not in ref_sig and 'input,' not in ref_sig:
                assert str(inspect.signature(fn)) == ref_sig, name


def test_store_matches_reference_casts(api):
    store = fewbit.functional.activations.store
    assert len(store) == api['store_len'] > 0
    assert sorted(f'{k[0]}{k[1]:02d}' for k, _ in store.items()) == api['store_keys']
    assert store.get('gelu', 3) is not None
    for key, entry in api['tables_cast'].items():
        name, bits, dt = key[:-6] if key.endswith('_bf16') else key.rsplit('_', 1)[0][:-2], None, key.rsplit('_', 1)[1]
        name, bits = key.rsplit('_', 1)[0][:-2], int(key.rsplit('_', 1)[0][-2:])
        b, l = store.get(name, bits, 'cpu', DTYPES[dt])
        view = torch.int32 if dt == 'f32' else torch.int16
        npdt = np.uint32 if dt == 'f32' else np.uint16
        assert b.view(view).numpy().view(npdt).tolist() == entry['borders'], key
        assert l.view(view).numpy().view(npdt).tolist() == entry['levels'], key
    with pytest.raises(KeyError):
        store.get('gelu', 7)


def test_error_behaviour_matches_reference(api):
    x = torch.zeros(4)
    assert api['errors'] == {'bits_and_custom': 'ValueError', 'unknown_bits': 'KeyError'}
    with pytest.raises(ValueError):
        fewbit.functional.gelu(x, bits=3, borders=torch.zeros(3), values=torch.zeros(4))
    with pytest.raises(KeyError):
        fewbit.functional.gelu(x, bits=7)
    with pytest.raises(ValueError):   # size mismatch, fewbit/functional/activations.py:112-114
        fewbit.functional.gelu(x, borders=torch.tensor([-100., 0., 1., 100.]), values=torch.zeros(5))
    with pytest.raises(ValueError):   # a shift is the origin of a symmetry: meaningless without `parity`
        fewbit.functional.stepwise(x, torch.zeros(1), torch.zeros(2), shift=(1.0, 0.0))
    with pytest.raises(TypeError):
        fewbit.functional.threshold(x)          # threshold and value are required
    with pytest.raises(ValueError):
        fewbit.Stepwise(torch.zeros(2), torch.zeros(5))


@pytest.mark.parametrize('name', fewbit.functional.CONTINOUS)
@pytest.mark.parametrize('bits', (1, 2, 3, 4))
def test_host_path_gradient_is_the_table(name, bits):
    """Host tensors: forward equals torch's, gradient equals levels[searchsorted(borders, x)] * gy exactly."""
    fn = getattr(fewbit.functional, name)
    ref = getattr(torch, name) if name in ('sigmoid', 'tanh') else getattr(F, name)
    x = torch.linspace(-5, 5, 101, requires_grad=True)
    gy = torch.linspace(0.5, 1.5, 101)
    y = fn(x, bits=bits)
    y.backward(gy)
    assert torch.equal(y.detach(), ref(x.detach()))
    borders, levels = fewbit.functional.store.get(name, bits)
    codes = torch.searchsorted(borders[1:-1].contiguous(), x.detach())
    assert torch.equal(x.grad, levels[codes] * gy)


def test_host_path_matches_oracle_bitwise():
    import oracle
    x = (torch.randn(1001, generator=torch.Generator().manual_seed(3)) * 2).to(torch.bfloat16)
    gy = torch.randn(1001, generator=torch.Generator().manual_seed(4)).to(torch.bfloat16)
    xr = x.clone().requires_grad_()
    fewbit.functional.gelu(xr, bits=3).backward(gy)
    borders, levels = fewbit.functional.store.get('gelu', 3, 'cpu', torch.bfloat16)
    _, state, _ = oracle.quantize('gelu', x, borders[1:-1])
    assert_bit_equal(xr.grad, oracle.quantize_backward(gy, state, levels), 'host grad vs oracle')


@pytest.mark.parametrize('name,args', [('hardshrink', ()), ('hardshrink', (1.0,)), ('hardsigmoid', ()), ('hardtanh', ()),
                                       ('hardtanh', (-2.0, 2.0)), ('leaky_relu', ()), ('leaky_relu', (0.5,)),
                                       ('relu', ()), ('relu6', ()), ('softshrink', ()), ('softshrink', (1.0,)),
                                       ('threshold', (1.0, 3.0))])
def test_host_stepwise_functions(name, args):
    # the reference's TestStepwiseFunctions (fewbit/functional/activations_test.py:16-68) on host tensors
    x = torch.linspace(-5, 5, 101)
    p = x.clone().requires_grad_()
    q = x.clone().requires_grad_()
    ys = getattr(F, name)(p, *args)
    ys.backward(torch.ones_like(x))
    zs = getattr(fewbit.functional, name)(q.clone(), *args)
    zs.backward(torch.ones_like(x))
    assert torch.linalg.norm(zs - ys).item() < 1e-6 and torch.linalg.norm(p.grad - q.grad).item() < 1e-6


def test_modules_on_host_and_bits_kwarg():
    x = torch.linspace(-3, 3, 25)
    assert torch.equal(fewbit.ReLU()(x.clone()), F.relu(x))                   # SURVEY 2.2 defect 4 fixed
    assert torch.equal(fewbit.Hardtanh(-2.0, 2.0)(x.clone()), F.hardtanh(x, -2.0, 2.0))
    assert torch.equal(fewbit.LeakyReLU(0.2)(x.clone()), F.leaky_relu(x, 0.2))
    assert torch.equal(fewbit.GELU(bits=2)(x.clone()), F.gelu(x))
    assert torch.allclose(fewbit.Softplus(beta=2.0)(x.clone()), F.softplus(x, beta=2.0))
    b, l = fewbit.functional.store.get('tanh', 3)
    m = fewbit.Stepwise(b, l)
    assert m.borders.numel() == 7 and sorted(m.state_dict()) == ['borders', 'levels']
    xr = x.clone().requires_grad_()
    m(xr).sum().backward()
    assert torch.equal(xr.grad, l[torch.searchsorted(b[1:-1].contiguous(), x)])


def test_map_module_and_memory_hooks():
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.GELU(), torch.nn.Sequential(torch.nn.ReLU(), torch.nn.GELU()))
    seen = []

    def swap(mod, path):
        seen.append(path)
        return fewbit.GELU(bits=3) if isinstance(mod, torch.nn.GELU) else mod

    out = fewbit.map_module(net, swap)
    assert out is net and isinstance(net[1], fewbit.GELU) and isinstance(net[2][1], fewbit.GELU)
    assert seen[-1] == '/' and '/2/1' in seen
    only = fewbit.map_module(torch.nn.Sequential(torch.nn.GELU(), torch.nn.GELU()),
                             lambda m, p: fewbit.GELU(bits=2) if isinstance(m, torch.nn.GELU) else m, r'/0')
    assert isinstance(only[0], fewbit.GELU) and isinstance(only[1], torch.nn.GELU)
    with pytest.raises(ValueError):
        fewbit.map_module(net, lambda m, p: None)
    # fewbit/util_test.py: saved-tensor byte accounting
    x = torch.randn(3, 4, requires_grad=True)
    with fewbit.memory_usage_hooks() as usage:
        torch.relu(x).sum().backward()
    assert usage.forward == 3 * 4 * 4 and usage.value == usage.backward
    with fewbit.memory_usage_hooks() as usage:
        fewbit.functional.gelu(x, bits=3).sum().backward()
    assert usage.forward == 12 + 8 * 4        # one byte per element on the host path + the level table


def test_gpu_tensor_without_native_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(fewbit_amd, '_native_loaded', False)
    monkeypatch.setattr(fewbit_amd, '_native_error', 'simulated')
    with pytest.raises(RuntimeError, match='native library is not loaded'):
        fewbit_amd.functional._native_op('gelu')


def test_host_stepwise_parity_and_shift():
    """`Stepwise(borders, levels, parity, shift)` (declared by the reference, fewbit/modules/activations.py:97-134,
    implemented nowhere in it).  Even: level = l'[#{b' < |x - sx|}].  Odd about (sx, sy): the mirrored plain table.
    The host path must agree with the oracle's codes and with a direct restatement of the definition."""
    b = torch.tensor([0.5, 1.0, 2.0])
    l = torch.tensor([1.0, 0.6, 0.3, 0.1])
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(500, generator=g) * 2, torch.tensor([0.0, -0.0, 0.5, -0.5, 1.5, 2.5, -1.0, float('nan'),
                                                                      float('inf'), -float('inf')])])
    for sx in (0.0, 1.0, -0.25):
        xr = x.clone().requires_grad_()
        y = fewbit.Stepwise(b, l, parity=True, shift=(sx, 7.0))(xr * 1.0)
        y.backward(torch.ones_like(y))
        _, st, k = oracle.quantize('identity_fold', x, b, sx)
        codes = torch.from_numpy(oracle.inflate(st.numpy(), x.numel(), k)).long()
        assert k == 2 and torch.equal(xr.grad, l[codes])
        t = (x - sx).abs()
        want = torch.where(t > 2.0, 0.1, torch.where(t > 1.0, 0.3, torch.where(t > 0.5, 0.6, 1.0)))
        want[torch.isnan(x)] = 0.1
        assert torch.equal(xr.grad, want)
        assert torch.equal(y.detach()[~torch.isnan(x)], x[~torch.isnan(x)])
        # odd about (sx, 0.5): right of sx the table itself, left of it 2*sy - l'
        xr = x.clone().requires_grad_()
        fewbit.functional.stepwise(xr * 1.0, b, l, parity=False, shift=(sx, 0.5)).sum().backward()
        right = x > sx
        # a value ON a mirrored border belongs to the bucket further left (plain-table rule "b < x"), i.e. closed
        # intervals in t on the left side
        left = torch.where(t >= 2.0, 0.1, torch.where(t >= 1.0, 0.3, torch.where(t >= 0.5, 0.6, 1.0)))
        want_odd = torch.where(right, want, 1.0 - left)
        want_odd[torch.isnan(x)] = 0.1
        assert torch.equal(xr.grad, want_odd)


def test_graph_accounting_like_reference_util_test():
    """fewbit/util_test.py restated: traverse/teniter/estimate_memory_usage/memory_usage_hooks on two tiny MLPs."""
    from fewbit.util import convert_linear, estimate_memory_usage, memory_usage_hooks, teniter, traverse
    torch.manual_seed(0)
    m1 = torch.nn.Sequential(torch.nn.Linear(8, 4), torch.nn.Linear(4, 1))
    m2 = torch.nn.Sequential(torch.nn.Linear(8, 4), torch.nn.ReLU(), torch.nn.Linear(4, 1))
    xs = torch.randn(3, 8)
    traverse(m1(xs.requires_grad_()), lambda node, ten, saved: None)
    n1 = len(list(teniter(m1(xs.requires_grad_()), False, True)))
    n2 = len(list(teniter(m2(xs.requires_grad_()), False, True)))
    assert n1 + 1 == n2
    assert estimate_memory_usage(m1(xs.requires_grad_())) == 4 * (3 * 8 + 4 * 8 + 4 + 1 * 4 + 1)
    size = 3 * 4 * 4                                              # the ReLU output, saved for its backward
    assert estimate_memory_usage(m2(xs.requires_grad_()), True) - estimate_memory_usage(m1(xs.requires_grad_()), True) == size
    with memory_usage_hooks() as lhs:
        m1(torch.randn(3, 8).requires_grad_())
    with memory_usage_hooks() as rhs:
        x2 = torch.randn(3, 8)
        m2(x2.requires_grad_()).backward(torch.ones(3, 1))
    assert rhs.value - lhs.value == size
    # Linear -> RandomizedLinear through map_module + convert_linear, parameters shared
    net = fewbit.map_module(m2, lambda m, p: convert_linear(m, fewbit.RandomizedLinear, proj_dim_ratio=0.5))
    assert isinstance(net[0], fewbit.RandomizedLinear) and isinstance(net[1], torch.nn.ReLU)
    assert net(torch.randn(6, 8)).shape == (6, 1)


def test_host_path_refuses_tables_its_byte_codes_cannot_hold():
    """codes are one byte each on the host path: 257 levels (or an odd-parity table that mirrors beyond 256) must
    raise instead of wrapping silently (the GPU operators have the same limits, torch_ops.cpp)."""
    x = torch.randn(64, requires_grad=True)
    b = torch.linspace(-3, 3, 258)
    with pytest.raises(ValueError):
        fewbit.functional.gelu(x, borders=b, values=torch.rand(257))
    y = fewbit.functional.gelu(x, borders=b[:257], values=torch.rand(256))           # 256 levels still fit
    y.sum().backward()
    half_b, half_l = torch.linspace(0.01, 3, 128), torch.rand(129)
    with pytest.raises(ValueError):
        fewbit.functional.stepwise(x, half_b, half_l, parity=False)
    half_b, half_l = torch.linspace(0.01, 3, 127), torch.rand(128)
    xx = torch.randn(500, requires_grad=True)
    fewbit.functional.stepwise(xx, half_b, half_l, parity=False, shift=(0.0, 0.0)).sum().backward()
    code = torch.searchsorted(half_b, xx.detach().abs())
    want = torch.where(xx.detach() > 0, half_l[code], -half_l[code])
    assert torch.equal(xx.grad, want)


def test_reference_submodule_paths_resolve():
    """Every import path of the reference package resolves on the alias (users import e.g.
    ``from fewbit.functional.activations import store`` or ``from fewbit.compat import removeprefix``)."""
    import importlib
    for path in ('fewbit.functional.activations', 'fewbit.functional.linear', 'fewbit.functional.variance',
                 'fewbit.modules.activations', 'fewbit.modules.linear', 'fewbit.modules.variance', 'fewbit.util', 'fewbit.fft',
                 'fewbit.approx', 'fewbit.cli', 'fewbit.compat'):
        importlib.import_module(path)
    from fewbit.compat import removeprefix
    from fewbit.functional.activations import store
    from fewbit.modules.variance import VarianceEstimator  # noqa: F401
    assert removeprefix('module.weight', 'module.') == 'weight' and removeprefix('abc', 'x') == 'abc'
    assert ('gelu', 3) in store


def test_inference_mode_first_then_training():
    """A table cast that is first made under torch.inference_mode() must still be an ordinary tensor: it is cached and a
    later training call saves it for backward (advisor finding, round 2)."""
    from fewbit_amd.store import StepwiseStore, BUILTIN_TABLES
    fresh = StepwiseStore().load(BUILTIN_TABLES)
    with torch.inference_mode():
        b, l = fresh.get('gelu', 3, 'cpu', torch.bfloat16)
        bi, li = fresh.get_inner('gelu', 3, torch.device('cpu'), torch.bfloat16)
    assert not any(t.is_inference() for t in (b, l, bi, li))
    # end to end on the shared store: module, functional and raw operator, each first under inference_mode
    x = torch.randn(257)
    module = fewbit.Mish(bits=2)
    with torch.inference_mode():
        module(x.clone())
        fewbit.functional.softsign(x.clone(), bits=4)
        inner, levels = fewbit.functional.store.get_inner('selu', 3, torch.device('cpu'), torch.float32)
        if fewbit_amd.native_loaded():
            torch.ops.fewbit.selu(x.clone(), inner, levels)
    for call in (module, lambda t: fewbit.functional.softsign(t, bits=4)):
        xg = x.clone().requires_grad_()
        call(xg).sum().backward()
        assert xg.grad is not None and torch.isfinite(xg.grad).all()
    if fewbit_amd.native_loaded():
        xg = x.clone().requires_grad_()
        torch.ops.fewbit.selu(xg.clone(), inner, levels).sum().backward()
        assert torch.equal(xg.grad, levels[torch.searchsorted(inner, x)])
