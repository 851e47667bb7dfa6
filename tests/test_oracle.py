"""Pins the CPU oracle (oracle/) against the reference: its own known-answer vector, golden vectors
produced by running the reference (tests/golden/gen_golden.py) and, when present, the reference
codec compiled where it lies (oracle/_ref).  CPU only."""
import numpy as np
import pytest
import torch

import oracle
from helpers import DTYPES, GOLDEN, assert_bit_equal, forward_value_ok, from_raw, ulp_distance


@pytest.fixture(scope='module')
def qref():
    with np.load(GOLDEN / 'quantize_ref.npz') as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='module')
def cref():
    with np.load(GOLDEN / 'codec_ref.npz') as z:
        return {k: z[k] for k in z.files}


def test_reference_known_answer_vector():
    # fewbit/cpu/codec_test.cc:9-22: {0,1,4,7} @ 3 bit round-trips; bytes probed from the reference
    packed = oracle.deflate([0, 1, 4, 7], 3)
    assert packed.tolist() == [0x08, 0x0f]
    assert oracle.inflate(packed, 4, 3).tolist() == [0, 1, 4, 7]
    # SURVEY section 0 known answers (probed from the reference codec)
    assert oracle.deflate([0, 0, 0, 1, 1, 3, 6, 6, 7, 7, 7], 3).tolist() == [0, 146, 217, 255, 1]
    assert oracle.deflate([1, 2, 1, 0, 0, 1, 3, 3, 1, 3, 3, 1, 2, 3, 0, 2], 2).tolist() == [25, 244, 125, 142]
    assert oracle.deflate([13, 9, 11, 10, 3, 14, 8, 14, 5, 4, 12, 14, 11, 5, 9, 11],
                          4).tolist() == [157, 171, 227, 232, 69, 236, 91, 185]


@pytest.mark.parametrize('k', range(1, 9))
def test_codec_golden(cref, k):
    for n in (1, 5, 8, 11, 64, 256, 1001):
        codes = cref[f'k{k}_n{n}_codes'].astype(np.int32)
        want = cref[f'k{k}_n{n}_bytes']
        got = oracle.deflate(codes, k)
        assert got.tolist() == want.tolist(), (k, n)
        assert oracle.inflate(want, n, k).tolist() == codes.tolist()


@pytest.mark.parametrize('k', range(1, 9))
def test_codec_randomized_like_reference(k):
    # fewbit/cpu/codec_test.cc:24-51: widths 1..8, 256 random codes round-trip
    rng = np.random.default_rng(42)
    codes = rng.integers(0, 1 << k, 256).astype(np.int32)
    packed = oracle.deflate(codes, k)
    assert packed.size == int(np.ceil(k / 8. * 256))
    assert (oracle.inflate(packed, 256, k) == codes).all()


def test_codec_against_compiled_reference():
    R = oracle.ref_codec()
    if R is None:
        pytest.skip('oracle/_ref not built (reference tree absent)')
    rng = np.random.default_rng(3)
    for k in range(1, 9):
        for n in (1, 2, 3, 7, 8, 9, 15, 16, 17, 100, 4093):
            codes = rng.integers(0, 1 << k, n).astype(np.int32)
            mine = oracle.deflate(codes, k)
            ref = np.zeros(mine.size + 2, np.uint8)
            R.ref_deflate_u8(codes.ctypes.data, n, ref.ctypes.data, k)
            assert (mine == ref[:mine.size]).all(), (k, n)
            back = np.zeros(n, np.int32)
            R.ref_inflate_u8(back.ctypes.data, n, mine.ctypes.data, k)
            assert (back == codes).all()


@pytest.mark.parametrize('dt', list(DTYPES))
def test_software_conversions_match_torch(dt):
    dtype = DTYPES[dt]
    g = torch.Generator().manual_seed(0)
    for scale in (1.0, 1e-6, 1e-3, 300.0, 7e4):
        x = torch.randn(20000, generator=g) * scale
        assert_bit_equal(oracle.convert(x, dtype), x.to(dtype), f'f32->{dt} scale {scale}')
    if dtype != torch.float32:
        h = torch.arange(-32768, 32768, dtype=torch.int32).to(torch.int16).view(dtype)
        assert_bit_equal(oracle.convert(h, torch.float32), h.float(), f'{dt}->f32')


@pytest.mark.parametrize('k', (2, 3, 4))
@pytest.mark.parametrize('dt', list(DTYPES))
def test_quantize_matches_reference_run(qref, k, dt):
    """codes/state/gx bit-exact with torch.ops.fewbit.quantize(_backward) of the reference."""
    dtype = DTYPES[dt]
    borders = from_raw(qref[f'gelu{k:02d}_{dt}_borders'], dtype)
    levels = from_raw(qref[f'gelu{k:02d}_{dt}_levels'], dtype)
    for n in (1, 7, 8, 9, 64, 65, 257, 1001):
        key = f'gelu{k:02d}_{dt}_{n}'
        x, gy = from_raw(qref[key + '_x'], dtype), from_raw(qref[key + '_gy'], dtype)
        y, state, kk = oracle.quantize('gelu', x, borders)
        assert kk == k
        want_state = torch.from_numpy(qref[key + '_state'])
        assert want_state.numel() == oracle.state_nbytes(n, k)
        assert state.numel() == oracle.state_nbytes_padded(n, k)
        assert_bit_equal(state[:want_state.numel()], want_state, key + ' state')
        assert not state[want_state.numel():].any(), 'padding must be zero'
        gx = oracle.quantize_backward(gy, state, levels)
        assert_bit_equal(gx, from_raw(qref[key + '_gx'], dtype), key + ' gx')
        # forward values come from ATen (third-party arithmetic); tolerance: helpers.forward_value_ok
        y_ref = from_raw(qref[key + '_y'], dtype)
        ok = forward_value_ok(x, y, y_ref)
        assert ok.all(), (key, x[~ok][:5], y[~ok][:5], y_ref[~ok][:5])


@pytest.mark.parametrize('name,k,dt,n', [('silu', 2, 'f16', 1001), ('silu', 4, 'f16', 1001), ('tanh', 3, 'f32', 257)])
def test_other_tables_match_reference_run(qref, name, k, dt, n):
    dtype = DTYPES[dt]
    key = f'{name}{k:02d}_{dt}_{n}'
    borders = from_raw(qref[f'{name}{k:02d}_{dt}_borders'], dtype)
    levels = from_raw(qref[f'{name}{k:02d}_{dt}_levels'], dtype)
    x, gy = from_raw(qref[key + '_x'], dtype), from_raw(qref[key + '_gy'], dtype)
    _, state, kk = oracle.quantize(name, x, borders)
    assert kk == k
    want = torch.from_numpy(qref[key + '_state'])
    assert_bit_equal(state[:want.numel()], want, key + ' state')
    assert_bit_equal(oracle.quantize_backward(gy, state, levels), from_raw(qref[key + '_gx'], dtype), key + ' gx')


def test_searchsorted_rules():
    # x == border -> lower bucket; -0 == 0; +inf -> last; -inf -> 0; NaN -> last (either sign)
    b = torch.tensor([-1.0, 0.0, 2.0])
    x = torch.tensor([-1.0, 0.0, -0.0, 2.0, float('inf'), -float('inf'), float('nan'), 1e-45, 3.0])
    neg_nan = torch.tensor([-1], dtype=torch.int32).view(torch.float32)
    x = torch.cat([x, neg_nan])
    got = oracle.searchsorted(x, b).tolist()
    assert got == [0, 1, 1, 2, 3, 0, 3, 2, 3, 3]
    assert got == torch.searchsorted(b, x, out_int32=True).tolist()      # ATen's own rule, same answers
    for dtype in (torch.bfloat16, torch.float16):
        g = torch.Generator().manual_seed(1)
        xs = (torch.randn(5000, generator=g) * 2).to(dtype)
        bs = torch.tensor([-2.4, -0.7, -0.3, 1e-4, 0.3, 0.7, 2.4]).to(dtype)
        assert oracle.searchsorted(xs, bs).tolist() == torch.searchsorted(bs, xs, out_int32=True).tolist()


@pytest.mark.parametrize('dt', list(DTYPES))
def test_relu_1bit_golden(cref, dt):
    """1-bit path: bit rule of fewbit/cuda/codec.cu:412-425 packed by the reference's Deflate(...,1)."""
    dtype = DTYPES[dt]
    x = from_raw(cref[f'relu01_{dt}_x'], dtype)
    y, state = oracle.stepwise1_forward('relu', x)
    # NaN: `x <= 0` is false -> bit 1, value NaN (ATen relu propagates NaN)
    assert_bit_equal(state, torch.from_numpy(cref[f'relu01_{dt}_state']), 'relu state')
    # value equality: the reference kernel writes +0.0 for x <= 0 (fewbit/cuda/codec.cu:417-418) where
    # ATen's relu keeps -0.0; the sign of zero is not part of parity
    y_ref = from_raw(cref[f'relu01_{dt}_y'], dtype)
    assert ((y == y_ref) | (torch.isnan(y) & torch.isnan(y_ref))).all()
    gy = from_raw(cref[f'relu01_{dt}_gy'], dtype)
    gx = oracle.stepwise1_backward('relu', gy, state)
    want = torch.where((x.float() > 0) | torch.isnan(x.float()), gy.float(), gy.float() * 0).to(dtype)
    assert_bit_equal(gx, want, 'relu gx')


def _torch_fn(name):
    import torch.nn.functional as F
    return getattr(torch, name) if name in ('sigmoid', 'tanh') else getattr(F, name)


@pytest.mark.parametrize('name', [f for f in oracle.CONTINUOUS if not f.startswith('identity')])
def test_continuous_forward_tracks_torch(name):
    """Reference test bar (fewbit/functional/activations_test.py:81-89): ||fewbit(x) - F(x)||_2 <= 1e-6
    on linspace(-5, 5, 101); here additionally max 2 fp32 steps relative to max(|y|, tiny)."""
    x = torch.linspace(-5, 5, 101)
    y = oracle.activation(name, x, *({'celu': (1.0,), 'elu': (1.0,), 'softplus': (1.0, 20.0)}.get(name, ())))
    ref = _torch_fn(name)(x)
    # gelu: ATen-CPU fp32 (MKL vsCdfNorm) and the erf formula differ by cancellation noise for x < -2, hence 4e-6
    assert torch.linalg.norm(y - ref).item() <= (4e-6 if name == 'gelu' else 1e-6)
    for dtype in (torch.bfloat16, torch.float16):
        xs = x.to(dtype)
        ys = oracle.activation(name, xs, *({'celu': (1.0,), 'elu': (1.0,), 'softplus': (1.0, 20.0)}.get(name, ())))
        # 16-bit I/O is this build's extension (the reference's GPU path is fp32 only): defined as fp32
        # evaluation of the 16-bit input, rounded once (ATen's composite ops, e.g. tanhshrink, round twice)
        assert forward_value_ok(xs, ys, _torch_fn(name)(xs.float()).to(dtype)).all(), (name, dtype)


@pytest.mark.parametrize('name,args', [('hardshrink', ()), ('hardshrink', (1.0,)), ('hardsigmoid', ()),
                                       ('hardtanh', ()), ('hardtanh', (-2.0, 2.0)), ('leaky_relu', ()),
                                       ('leaky_relu', (0.5,)), ('relu', ()), ('relu6', ()), ('softshrink', ()),
                                       ('softshrink', (1.0,)), ('threshold', (1.0, 3.0))])
def test_stepwise1_tracks_torch(name, args):
    """fewbit/functional/activations_test.py:16-68: value and gradient equal torch's on linspace(-5,5,101)
    with gs = ones (to 6 places there; exact here except hardsigmoid's division)."""
    import torch.nn.functional as F
    defaults = {'hardshrink': (0.5,), 'hardtanh': (-1.0, 1.0), 'leaky_relu': (0.01,), 'softshrink': (0.5,)}
    p = args or defaults.get(name, ())
    x = torch.linspace(-5, 5, 101).requires_grad_()
    ref = getattr(F, name)(x, *p)
    ref.backward(torch.ones_like(ref))
    y, state = oracle.stepwise1_forward(name, x.detach(), *p)
    assert torch.linalg.norm(y - ref.detach()).item() < 1e-6
    # only leaky_relu's backward takes a parameter (its slope)
    gx = oracle.stepwise1_backward(name, torch.ones(101), state, *(p[:1] if name == 'leaky_relu' else ()))
    assert torch.linalg.norm(gx - x.grad).item() < 1e-6


@pytest.mark.parametrize('dtype', (torch.float32, torch.bfloat16, torch.float16))
def test_folded_key_is_the_fp32_distance_from_the_shift(dtype):
    """FEWBIT_IDENTITY_FOLD (include/fewbit_hip.h): code = #{b' < |x - sx|}, the subtraction in fp32.  No reference
    behaviour exists (declared only, fewbit/fewbit.cc:37): pinned against a numpy restatement of the definition."""
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(4099, generator=g) * 2).to(dtype)
    x[:4] = torch.tensor([float('nan'), float('inf'), -float('inf'), -0.0]).to(dtype)
    b = torch.tensor([0.25, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0]).to(dtype)
    for sx in (0.0, 0.3, -1.0):
        y, st, k = oracle.quantize('identity_fold', x, b, sx)
        codes = oracle.inflate(st.numpy(), x.numel(), k)
        key = np.abs(x.float().numpy() - np.float32(sx))
        want = np.searchsorted(b.float().numpy(), key, side='left')
        want[np.isnan(key)] = b.numel()
        assert k == 3 and np.array_equal(codes, want)
        iv = torch.int32 if dtype == torch.float32 else torch.int16
        assert torch.equal(y[1:].view(iv), x[1:].view(iv)) and torch.isnan(y[0])          # identity forward


def test_reference_smoke_vectors_pin_each_other():
    """fewbit/cuda/codec_test.cu: the 16 GELU inputs of TestGelu land, with the built-in 3-bit table, on the 16 codes of
    TestCodecBlock; their packed form round-trips and the backward with unit gradients returns the table levels."""
    from helpers import REF_SMOKE_CODES, REF_SMOKE_GELU_INPUTS
    from fewbit_amd.store import store
    x = torch.tensor(REF_SMOKE_GELU_INPUTS, dtype=torch.float32)
    borders, levels = store.get('gelu', 3, 'cpu', torch.float32)
    y, state, k = oracle.quantize('gelu', x, borders[1:-1])
    assert k == 3 and state.numel() == 6
    assert oracle.inflate(state.numpy(), 16, 3).tolist() == list(REF_SMOKE_CODES)
    assert np.array_equal(oracle.deflate(np.array(REF_SMOKE_CODES, dtype=np.int32), 3), state.numpy())
    assert torch.equal(oracle.quantize_backward(torch.ones(16), state, levels), levels[torch.tensor(REF_SMOKE_CODES)])
    assert y[4].item() == pytest.approx(999.9) and y[15].item() == pytest.approx(999.9)


def test_oracle_under_address_and_ub_sanitizers():
    """the CPU restatement is run under ASan/UBSan (build container only, never on the GPU box):
    `make -C oracle asan-test` re-runs this file and tests/test_host_ops.py against liboracle_asan.so."""
    import os
    import shutil
    import subprocess
    if os.environ.get('FEWBIT_ORACLE_LIB'):
        pytest.skip('already inside the sanitizer run')
    if not shutil.which('gcc') or not os.path.exists(subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True,
                                                                     text=True).stdout.strip()):
        pytest.skip('no libasan in this image')
    from helpers import ROOT
    r = subprocess.run(['make', '-s', '-C', str(ROOT / 'oracle'), 'asan-test'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert ' passed' in r.stdout and 'ERROR: AddressSanitizer' not in r.stdout + r.stderr and 'runtime error' not in r.stderr
