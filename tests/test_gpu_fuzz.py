"""Seeded differential fuzz of the C-ABI against the oracle: random sizes (around tile boundaries and ragged ends),
dtypes, tables of 2..256 levels (full and ragged bit widths), element/byte offsets of every pointer, in-place and
out-of-place, several functors -- packed bytes and gradients must be bit-identical, forward values within the
stated tolerance.  Every kernel family is reached: register search, pattern table (via a lowered threshold in a
subprocess-free way: large n), LDS search, wide backward, 1-bit family."""
import numpy as np
import pytest
import torch

import oracle
from fewbit_amd import cabi
from helpers import DTYPES, assert_bit_equal, forward_value_ok, ulp_distance

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

SIZES = (1, 7, 8, 9, 63, 64, 65, 511, 512, 513, 520, 1023, 4095, 4096, 4097, 8191, 12345, 65536, 65543, 100003)
FUNCTORS = ('gelu', 'silu', 'identity', 'tanh', 'softplus', 'identity_fold')
PARAMS = {'softplus': (1.0, 20.0), 'identity_fold': (0.375, 0.0)}


def random_table(g, nlev, dtype, fold):
    lo = 0.05 if fold else -3.0
    inner = torch.unique((torch.rand(nlev - 1, generator=g) * (3.0 - lo) + lo).to(dtype))
    levels = (torch.rand(inner.numel() + 1, generator=g) * 2 - 0.5).to(dtype)
    return inner, levels


def offset_copy(t, off_elems, extra=16):
    """the same values in a fresh device buffer that starts `off_elems` elements past an aligned address"""
    buf = torch.empty(t.numel() + off_elems + extra, dtype=t.dtype, device=DEV)
    view = buf[off_elems:off_elems + t.numel()]
    view.copy_(t)
    return view


TUNE_KEYS = ('waves_per_cu', 'chunk', 'lut_chunk', 'u_fwd', 'u_bwd', 'u_step1')


def random_launch_shape(rng):
    """groups per lane per stage, resident blocks per CU and resident / chunked shape picked per case (fewbit_hip_tune):
    every kernel instantiation the size policy can choose is reached at the small sizes of this file too"""
    u = int(rng.choice([-1, 1, 2]))
    cabi.tune(u_fwd=u, u_bwd=int(rng.choice([-1, 1, 2])), u_step1=int(rng.choice([-1, 1, 2])),
              waves_per_cu=int(rng.choice([-1, 8, 16, 32])), chunk=int(rng.choice([-1, 0, 1, 3])), lut_chunk=int(rng.choice([-1, 0, 2])))


@pytest.fixture(autouse=True)
def _restore_launch_policy():
    yield
    cabi.tune(**{k: -1 for k in TUNE_KEYS})


# FEWBIT_FUZZ_SEEDS=N widens the sweep (a one-off soak run; the default keeps the suite short)
N_SEEDS = int(__import__('os').environ.get('FEWBIT_FUZZ_SEEDS', '12'))


@pytest.mark.parametrize('seed', range(N_SEEDS))
def test_fuzz_quantize_paths(seed):
    g = torch.Generator().manual_seed(1000 + seed)
    rng = np.random.default_rng(seed)
    for _ in range(14):
        dt = ('f32', 'f16', 'bf16')[rng.integers(3)]
        dtype = DTYPES[dt]
        fn = FUNCTORS[rng.integers(len(FUNCTORS))]
        nlev = int(rng.choice([2, 3, 4, 5, 7, 8, 11, 16, 17, 31, 32, 33, 64, 100, 128, 255, 256]))
        n = int(rng.choice(SIZES)) + int(rng.integers(0, 3)) * 8 * 64
        inner, levels = random_table(g, nlev, dtype, fn == 'identity_fold')
        k = oracle.bitwidth(inner.numel() + 1)
        x = (torch.randn(n, generator=g) * 2).to(dtype)
        if n >= 6:
            x[:6] = torch.tensor([float('nan'), float('inf'), -float('inf'), 0.0, -0.0, inner[0].item()]).to(dtype)
        gy = torch.randn(n, generator=g).to(dtype)
        p = PARAMS.get(fn, (0.0, 0.0))
        y_o, s_o, _ = oracle.quantize(fn, x, inner, *p)
        gx_o = oracle.quantize_backward(gy, s_o, levels)
        xo, so, go = (int(rng.integers(0, 8)) for _ in range(3))
        xd = offset_copy(x, xo)
        nbytes = cabi.state_nbytes(n, k)
        sbuf = torch.zeros(nbytes + so + 16, dtype=torch.uint8, device=DEV)
        st = sbuf[so:so + nbytes]
        inplace = bool(rng.integers(2))
        out = xd if inplace else offset_copy(torch.zeros_like(x), int(rng.integers(0, 8)))
        tag = f'seed={seed} {fn} {dt} nlev={nlev} n={n} offs=({xo},{so},{go}) inplace={inplace}'
        random_launch_shape(rng)
        y, _ = cabi.quantize_forward(fn, xd, inner.to(DEV), *p, out=out, state=st)
        assert_bit_equal(st.cpu(), s_o, tag + ' state')
        assert sbuf[:so].sum().item() == 0 and sbuf[so + nbytes:].sum().item() == 0, tag + ' wrote outside the state'
        fin = torch.isfinite(x)
        if fn == 'gelu':
            assert forward_value_ok(x, y.cpu(), y_o).all(), tag + ' y'
        else:   # libm (oracle) vs ocml / fast class (device): a few fp32 steps, or 1 step of a 16-bit output
            steps = ulp_distance(y.cpu()[fin], y_o[fin])
            diff = (y.cpu()[fin].double() - y_o[fin].double()).abs()
            assert ((steps <= (4 if dt == 'f32' else 1)) | (diff <= 1e-6)).all(), tag + ' y'
        gyd = offset_copy(gy, go)
        gx = cabi.quantize_backward(gyd, st, levels.to(DEV), out=gyd if inplace else None)
        assert_bit_equal(gx.cpu(), gx_o, tag + ' gx')


@pytest.mark.parametrize('seed', range(max(4, N_SEEDS // 3)))
def test_fuzz_one_bit_family(seed):
    g = torch.Generator().manual_seed(2000 + seed)
    rng = np.random.default_rng(100 + seed)
    cases = (('relu', ()), ('relu6', ()), ('leaky_relu', (0.1, )), ('hardtanh', (-0.5, 1.5)), ('hardshrink', (0.7, )),
             ('softshrink', (0.3, )), ('hardsigmoid', ()), ('threshold', (0.25, -1.0)))
    for _ in range(16):
        dt = ('f32', 'f16', 'bf16')[rng.integers(3)]
        dtype = DTYPES[dt]
        name, p = cases[rng.integers(len(cases))]
        n = int(rng.choice(SIZES)) + int(rng.integers(0, 3)) * 8 * 64
        x = (torch.randn(n, generator=g) * 2).to(dtype)
        gy = torch.randn(n, generator=g).to(dtype)
        y_o, s_o = oracle.stepwise1_forward(name, x, *p)
        gx_o = oracle.stepwise1_backward(name, gy, s_o, *(p[:1]))
        xo, so = int(rng.integers(0, 8)), int(rng.integers(0, 8))
        xd = offset_copy(x, xo)
        sbuf = torch.zeros(s_o.numel() + so + 16, dtype=torch.uint8, device=DEV)
        st = sbuf[so:so + s_o.numel()]
        tag = f'seed={seed} {name} {dt} n={n} offs=({xo},{so})'
        random_launch_shape(rng)
        y, _ = cabi.stepwise1_forward(name, xd, *p, state=st)
        assert_bit_equal(st.cpu(), s_o, tag + ' state')
        assert sbuf[:so].sum().item() == 0 and sbuf[so + s_o.numel():].sum().item() == 0, tag
        assert torch.equal(y.cpu().float(), y_o.float()), tag + ' y'          # value compare: relu(-0) sign
        gx = cabi.stepwise1_backward(name, offset_copy(gy, xo), st, *(p[:1]))
        assert_bit_equal(gx.cpu(), gx_o, tag + ' gx')


def test_fuzz_pattern_table_sizes():
    """Large tensors (pattern-table kernels, narrow and wide) with ragged ends and offsets."""
    g = torch.Generator().manual_seed(7)
    for dt, nlev, n, off in (('bf16', 8, (6 << 20) + 8 * 64 * 3 + 5, 3), ('f16', 6, (6 << 20) + 1, 1),
                             ('bf16', 40, (6 << 20) + 77, 5), ('f16', 256, (6 << 20) + 8 * 64, 2)):
        dtype = DTYPES[dt]
        inner, levels = random_table(g, nlev, dtype, False)
        k = oracle.bitwidth(inner.numel() + 1)
        x = (torch.randn(n, generator=g) * 2).to(dtype)
        x[:5] = torch.tensor([float('nan'), float('inf'), -float('inf'), 0.0, -0.0]).to(dtype)
        _, s_o, _ = oracle.quantize('gelu', x, inner)
        xd = offset_copy(x, off)
        nbytes = cabi.state_nbytes(n, k)
        sbuf = torch.zeros(nbytes + off + 16, dtype=torch.uint8, device=DEV)
        y, st = cabi.quantize_forward('gelu', xd, inner.to(DEV), state=sbuf[off:off + nbytes])
        assert_bit_equal(st.cpu(), s_o, f'{dt} nlev={nlev} n={n}')
        assert sbuf[:off].sum().item() == 0 and sbuf[off + nbytes:].sum().item() == 0


RAGGED_LARGE = (4_194_304 + 512 * 37 + 5, 8_650_003, 13_000_001, 17_825_792 + 8 * 63 + 7)


@pytest.mark.skipif('FEWBIT_SHAPE_WORKER' not in __import__('os').environ, reason='worker of test_every_launch_shape_at_ragged_large_sizes')
def test_shape_worker():
    """Tensors larger than one resident generation (so that a forced chunk setting really changes the launch shape), ragged
    ends, every kernel family; expectations are computed independently on the GPU with torch (bucketize / gather) and
    with the non-streaming codec kernel, plus an oracle window over the ragged end."""
    g = torch.Generator(device=DEV).manual_seed(5)
    for n in RAGGED_LARGE:
        for dtype in (torch.bfloat16, torch.float32):
            for name, k in (('gelu', 3), ('silu', 4)):
                from fewbit_amd.store import store
                borders, levels = store.get(name, k, DEV, dtype)
                inner = borders[1:-1].contiguous()
                x = (torch.randn(n, generator=g, device=DEV) * 1.5).to(dtype)
                gy = torch.randn(n, generator=g, device=DEV).to(dtype)
                y, state = cabi.quantize_forward(name, x, inner)
                codes = torch.bucketize(x.float(), inner.float(), out_int32=True)
                assert torch.equal(state, cabi.pack_codes(codes, k)), (n, dtype, name)
                gx = cabi.quantize_backward(gy, state, levels)
                want = (levels.float()[codes.long()] * gy.float()).to(dtype)
                assert_bit_equal(gx, want, f'gx {n} {dtype} {name}')
                tail = slice(n - 4099, n)                                   # oracle over the ragged end
                y_o, s_o, _ = oracle.quantize(name, x[tail].cpu(), inner.cpu())
                assert forward_value_ok(x[tail].cpu(), y[tail].cpu(), y_o).all()
                head = slice(0, 8 * 1000)
                y_o, s_o, _ = oracle.quantize(name, x[head].cpu(), inner.cpu())
                assert torch.equal(state[:k * 1000].cpu(), s_o)
        # a wide table (6 bits: the run-time-width kernels) and a custom ragged one (5 levels -> 3 bits, not full)
        for dtype, nlev in ((torch.float16, 33), (torch.float32, 40), (torch.bfloat16, 5)):
            k = cabi.bitwidth(nlev)
            inner = torch.sort(torch.randn(nlev - 1, generator=g, device=DEV))[0].to(dtype)
            inner = torch.unique(inner)
            levels = torch.randn(inner.numel() + 1, generator=g, device=DEV).to(dtype)
            k = cabi.bitwidth(levels.numel())
            x = (torch.randn(n, generator=g, device=DEV) * 1.5).to(dtype)
            gy = torch.randn(n, generator=g, device=DEV).to(dtype)
            y, state = cabi.quantize_forward('identity', x, inner)
            codes = torch.bucketize(x.float(), inner.float(), out_int32=True)
            assert torch.equal(state, cabi.pack_codes(codes, k)), (n, dtype, nlev)
            assert torch.equal(y.view(torch.uint8), x.view(torch.uint8))
            assert_bit_equal(cabi.quantize_backward(gy, state, levels), (levels.float()[codes.long()] * gy.float()).to(dtype))
        xh = torch.randn(n, generator=g, device=DEV).to(torch.bfloat16)
        yh, sth = cabi.stepwise1_forward('leaky_relu', xh, 0.25)
        assert torch.equal(sth, cabi.pack_codes((xh < 0).to(torch.int32), 1))
        assert torch.equal(yh, torch.nn.functional.leaky_relu(xh, 0.25))
        x = torch.randn(n, generator=g, device=DEV)
        gy = torch.randn(n, generator=g, device=DEV)
        y, st = cabi.stepwise1_forward('relu', x)
        assert torch.equal(y, torch.relu(x)) and torch.equal(st, cabi.pack_codes((x > 0).to(torch.int32), 1))
        assert torch.equal(cabi.stepwise1_backward('relu', gy, st), torch.where(x > 0, gy, torch.zeros_like(gy)))


@pytest.mark.parametrize('setting', ('0', '1', '2', '3', '5'))
def test_every_launch_shape_at_ragged_large_sizes(setting):
    """The streaming kernels have two launch shapes (resident round-robin / chunked, fewbit_kernels.hip `Span`) and a
    built-in policy that picks by tensor size.  Here the shape is forced through the tuning hooks (read once per process,
    hence the subprocess): 0 = always resident, T = chunks of T tiles per wave wherever the tensor has more tiles than
    resident waves -- every kernel family, ragged ends, partial last chunks."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, FEWBIT_HIP_CHUNK=setting, FEWBIT_HIP_LUT_CHUNK=setting, FEWBIT_SHAPE_WORKER='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', __file__, '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                        '-k', 'test_shape_worker'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and '1 passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
