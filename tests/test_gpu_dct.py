"""The sampled cosine transform on the GPU (fewbit_hip_sampled_dct, fewbit_amd/csrc/fewbit_dct.hip) against the REFERENCE'S OWN
`dct(x, dim=0, norm='ortho')[idx]` (fewbit/functional/linear.py:121-122 with fewbit/fft.py:10-43), run in the build container:
tests/golden/sampled_dct_ref.npz (generator: tests/golden/gen_linear_golden.py dct).  Floating point, tolerance stated per check:
the kernel computes in fp32 (four-step FFT, ~log2(rows) roundings), the reference call was float64, so

    fp32 input   |err| <= 3e-6 * max|y|                              (single-precision FFT of up to 65536 points)
    bf16 / fp16  |err| <= 2^-8 |y| / 2^-11 |y| + 3e-6 * max|y|       (one rounding of the fp32 result to the 16-bit dtype)
"""
import numpy as np
import pytest
import torch

import fewbit
from fewbit_amd import cabi
from helpers import GOLDEN

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
REL = {torch.float32: 0.0, torch.float16: 2.0**-11, torch.bfloat16: 2.0**-8}


@pytest.fixture(scope='module', params=('sampled_dct_ref.npz', 'sampled_dct_ref_3x.npz', 'sampled_dct_ref_5x.npz', 'sampled_dct_ref_big.npz'))
def ref(request):
    """the reference's own rows: 2^8 .. 2^16 rows, 3 x 2^8 .. 3 x 2^14 and 5 x 2^8 .. 5 x 2^13 (the radix-3 / radix-5 first stage of pass B), 2^17 and 2^18 (512-point tiles)"""
    with np.load(GOLDEN / request.param) as z:
        return {k: z[k].copy() for k in z.files}


def close(got, want, dtype):
    want = torch.as_tensor(want, dtype=torch.float64)
    err = (got.detach().cpu().double() - want).abs()
    tol = REL[dtype] * want.abs() + 3e-6 * float(want.abs().max())
    return bool((err <= tol).all()), float((err / tol.clamp_min(1e-300)).max())


@pytest.mark.parametrize('dtype', (torch.float32, torch.bfloat16, torch.float16))
def test_sampled_rows_equal_the_reference_run(ref, dtype):
    """9 shapes, 256 .. 65536 rows, 7 shapes, 768 .. 49152 = 3 x 2^k rows, 6 shapes, 1280 .. 40960 = 5 x 2^k rows, and 131072 / 262144 rows; ragged and odd feature counts, corner rows (0, rows/2, rows-1, the self-paired residue classes,
    duplicates): every dtype against the reference's float64 rows (the inputs are exact in all three dtypes)"""
    for i in range(int(ref['cases'])):
        x = torch.from_numpy(ref[f'case{i}_x_times_16'].astype(np.float32) / 16.0).to(dtype).to(DEV)
        idx = torch.from_numpy(ref[f'case{i}_idx']).to(DEV)
        y = cabi.sampled_dct(x, idx)
        assert y.shape == (idx.numel(), x.shape[1]) and y.dtype == dtype
        ok, worst = close(y, ref[f'case{i}_y'], dtype)
        assert ok, (i, tuple(x.shape), dtype, worst)
        # the layer-level entry takes the same path and gives the same rows
        from fewbit_amd import linear
        assert 'fewbit_hip_sampled_dct' in linear.sampled_transform_path('dct', x)


def test_every_row_of_the_transform_scale_strides_and_buffers():
    """p = rows (every k once, shuffled) against torch's float64 transform of the same data on the host; `scale`, a strided input
    (leading dimension > features), caller-provided out / workspace, and bit-identical repeats"""
    g = torch.Generator().manual_seed(12)
    for rows, features in ((1024, 130), (256, 64), (16384, 66), (32768, 34), (65536, 4), (768, 64), (1536, 130), (12288, 66), (49152, 4), (131072, 6), (262144, 2), (1280, 64), (20480, 66), (40960, 4)):
        wide = torch.randn(rows, features + 6, generator=g).to(DEV)
        x = wide[:, 3:3 + features]                                           # unit stride along the features, ld = features + 6
        assert not x.is_contiguous()
        idx = torch.randperm(rows, generator=g).to(DEV)
        want = fewbit.fft.dct(x.double().cpu(), dim=0, norm='ortho')[idx.cpu()] * 2.5
        ws = torch.empty(cabi.sampled_dct_workspace_bytes(rows, features, rows, torch.float32), dtype=torch.uint8, device=DEV)
        out = torch.full((rows, features), float('nan'), device=DEV)
        y = cabi.sampled_dct(x, idx, 2.5, out=out, workspace=ws)
        assert y.data_ptr() == out.data_ptr()
        ok, worst = close(y, want, torch.float32)
        assert ok, (rows, features, worst)
        assert torch.equal(cabi.sampled_dct(x, idx, 2.5), y)                  # deterministic, workspace contents do not matter
    assert torch.equal(wide[:, :3], wide[:, :3]) and not torch.isnan(wide).any()


def test_shapes_without_a_kernel_are_refused_by_name_and_keep_the_library_path():
    assert cabi.sampled_dct_workspace_bytes(16384, 768, 3276) == 12 * 16384 * 256 + 2048 + 8 * 3276
    assert cabi.sampled_dct_workspace_bytes(16384, 70, 1) == 2 * 16384 * 256 + 2048 + 16
    assert cabi.sampled_dct_workspace_bytes(12288, 768, 2457) == 12 * 12288 * 256 + 2048 + 19664
    for rows in (48, 128, 384, 3000, 1792, 640, 81920, 98304, 524288):
        assert cabi.sampled_dct_workspace_bytes(rows, 64, 10) == 0
        x = torch.randn(rows, 8, device=DEV)
        idx = torch.zeros(4, dtype=torch.int64, device=DEV)
        with pytest.raises(cabi.FewbitHipError, match='rows'):
            cabi.sampled_dct(x, idx)
        from fewbit_amd import linear
        assert 'torch.fft' in linear.sampled_transform_path('dct', x)
    with pytest.raises(cabi.FewbitHipError, match='int64'):
        cabi.sampled_dct(torch.randn(256, 8, device=DEV), torch.zeros(4, dtype=torch.int32, device=DEV))
    assert cabi.sampled_dct(torch.randn(256, 8, device=DEV), torch.zeros(0, dtype=torch.int64, device=DEV)).shape == (0, 8)


@pytest.mark.parametrize('batch,seq', ((4, 128), (2, 384), (10, 128)))
@pytest.mark.parametrize('dtype', (torch.float32, torch.bfloat16))
def test_the_dct_layer_on_the_kernel_equals_the_layer_on_torch_fft(dtype, batch, seq, monkeypatch):
    """linear_grp(matmul='dct') with 512, with 768 = 3 x 256 and with 1280 = 5 x 256 rows.  The native path samples rows(seed) (seed = one host draw from the generator); the torch.fft
    formulation is handed the SAME rows (cabi.sampled_rows of that seed), so the weight gradient of the native path equals that of the
    torch.fft formulation (which tests/test_gpu_linear.py pins to the reference's deterministic outputs); forward, input gradient and
    bias gradient are exact on both"""
    from fewbit_amd import linear
    g = torch.Generator().manual_seed(3)
    x = torch.randn(batch, seq, 40, generator=g).to(dtype).to(DEV)
    w = (torch.randn(24, 40, generator=g) * 0.3).to(dtype).to(DEV)
    b = torch.randn(24, generator=g).to(dtype).to(DEV)
    gy = torch.randn(batch, seq, 24, generator=g).to(dtype).to(DEV)
    assert 'fewbit_hip_sampled_dct' in linear.sampled_transform_path('dct', x.reshape(-1, 40))
    seed = int(torch.randint(0, 2**62, (), dtype=torch.int64, generator=torch.Generator().manual_seed(99)).item())     # linear._draw_seed
    grads = {}
    for native in (True, False):
        prev = linear.use_native_sketch(native)
        if not native:
            monkeypatch.setattr(linear, '_sampled_rows', lambda p, rows, like, gen: cabi.sampled_rows(seed, rows, p).to(like.device))
        try:
            xi, wi, bi = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
            gen = torch.Generator().manual_seed(99)
            y = fewbit.functional.linear_grp(xi, wi, bi, proj_dim_ratio=0.25, matmul='dct', generator=gen)
            y.backward(gy)
            grads[native] = (y.detach().float(), xi.grad.float(), bi.grad.float(), wi.grad.float())
        finally:
            linear.use_native_sketch(prev)
    for a, bb in zip(grads[True][:3], grads[False][:3]):
        assert torch.equal(a, bb)
    gw_n, gw_t = grads[True][3], grads[False][3]
    rel = float((gw_n - gw_t).abs().max() / gw_t.abs().max())
    assert rel <= (2e-5 if dtype == torch.float32 else 3e-2), rel          # bf16: both paths round the sampled rows to 8 bits, at different points


def test_the_seeded_transform_is_the_explicit_one_on_the_rows_of_the_seed():
    """fewbit_hip_sampled_dct_seeded(seed) == fewbit_hip_sampled_dct(idx = fewbit_hip_sampled_rows(seed)), bit for bit: every row split, the
    LDS list and its overflow (p >> the list: every workgroup walks the function group by group), the batched tail (p > 4096), p not a
    multiple of four, the seed by value and as a device word"""
    cases = [(256, 40, 9000), (256, 64, 1), (512, 66, 20001), (1024, 7, 300), (2048, 33, 2047), (4096, 64, 5000), (8192, 96, 1638),
             (16384, 768, 3276), (16384, 130, 16387), (32768, 64, 6553), (65536, 32, 13107),
             (131072, 64, 26214), (262144, 70, 52428), (262144, 2, 7),
             (1280, 40, 9000), (2560, 64, 5), (5120, 66, 1024), (10240, 96, 2048), (20480, 768, 4096), (40960, 32, 8192),
             (768, 40, 9000), (1536, 64, 3), (3072, 66, 20001), (6144, 96, 1229), (12288, 768, 2457), (24576, 64, 4915), (49152, 32, 9830)]
    for n, (rows, features, p) in enumerate(cases):
        dtype = (torch.bfloat16, torch.float32, torch.float16)[n % 3]
        x = torch.randn(rows, features, generator=torch.Generator().manual_seed(n)).to(dtype).to(DEV)
        seed = 0x9e3779b97f4a7c15 * (n + 1) & 0xffffffffffffffff
        idx = cabi.sampled_rows(seed, rows, p).to(DEV)
        want = cabi.sampled_dct(x, idx, 0.5)
        assert torch.equal(cabi.sampled_dct_seeded(x, p, seed, 0.5), want), (rows, features, p)
        # (ctypes hands the value over as an unsigned 64-bit word: the int64 tensor holds the same bits)
        word = torch.tensor([seed - (1 << 64) if seed >= 1 << 63 else seed], dtype=torch.int64, device=DEV)
        assert torch.equal(cabi.sampled_dct_seeded(x, p, word, 0.5), want), (rows, features, p)
        assert float(want.float().abs().max()) > 0


def test_a_captured_dct_layer_step_samples_fresh_rows_on_every_replay(monkeypatch):
    """The layer with matmul='dct' inside a hipGraph (the reference reads the RNG state back per call and cannot be captured): the recorded
    seed kernel derives the seed of replay r from (the host draw made at capture time, the device counter); the replayed weight gradient
    equals the explicit product on cabi.sampled_rows of that seed, backward meets forward's rows, replays differ"""
    from fewbit_amd import linear
    lin = fewbit.RandomizedLinear(64, 32, proj_dim=96, matmul='dct', bias=False, device=DEV)
    x = torch.randn(512, 64, device=DEV, requires_grad=True)
    wgt = torch.randn(512, 32, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.autograd.grad((lin(x) * wgt).sum(), lin.weight)
    torch.cuda.current_stream().wait_stream(side)
    base = 0x7654321
    monkeypatch.setattr(linear, '_draw_seed', lambda generator: base)
    counter = linear._replay_counter(torch.device(DEV))
    c0 = int(counter)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gw, = torch.autograd.grad((lin(x) * wgt).sum(), lin.weight)
    seen = []
    for r in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert int(counter) == c0 + r + 1
        idx = cabi.sampled_rows(cabi.mix_sketch_seed(base, c0 + r), 512, 96).to(DEV)
        want = cabi.sampled_dct(wgt, idx).T @ cabi.sampled_dct(x.detach(), idx, 512 / 96)
        assert torch.allclose(gw, want, rtol=1e-4, atol=1e-3), (r, float((gw - want).abs().max()))
        seen.append(gw.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


def _fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        rows = 1 << int(rng.integers(8, 17))
        if rng.integers(0, 3) == 0:
            rows = 3 << int(rng.integers(8, 15))
            if rng.integers(0, 2) == 0:
                rows = 5 << int(rng.integers(8, 14))
        features = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 63, 64, 65, 96, 127, 130, int(rng.integers(1, 200))]))
        if rows * features > (1 << 24):
            features = max(1, (1 << 24) // rows)
        p = int(rng.choice([1, 2, 17, 300, int(rng.integers(1, 3000)), rows]))
        dtype = (torch.float32, torch.bfloat16, torch.float16)[int(rng.integers(0, 3))]
        pad = int(rng.choice([0, 0, 3, 8]))
        yield rows, features, p, dtype, pad, float(rng.choice([1.0, 0.25, 5.0])), int(rng.integers(0, 2**31))


def test_sampled_dct_fuzz_against_float64_on_the_device():
    """40 random cases (FEWBIT_DCT_FUZZ_CASES=N widens the sweep): every supported row count (2^k, 3 x 2^k, 5 x 2^k), ragged / odd / tiny feature counts (edge
    tiles, a last complex column with no partner), p from 1 to rows, all three dtypes, strided inputs, scales; expected = the float64
    DCT-II of the same data computed by torch.fft on the device; per case also fewbit_hip_sampled_dct_seeded against the explicit call on
    fewbit_hip_sampled_rows of the same seed (bit-equal).  Plus the two list regimes of pass B forced on purpose: more samples of
    one residue pair than the LDS list holds (the group-by-group fallback), and p > 4096 (the batched tail of idx)."""
    import os
    n = int(os.environ.get('FEWBIT_DCT_FUZZ_CASES', '40'))
    cases = list(_fuzz_cases(n, 2026))
    cases += [(256, 40, 9000, torch.float32, 0, 1.0, 1), (512, 66, 20000, torch.bfloat16, 2, 1.0, 2), (16384, 34, 16384, torch.float32, 0, 1.0, 3),
              (768, 40, 9000, torch.float32, 0, 1.0, 1), (1536, 66, 20000, torch.bfloat16, 2, 1.0, 4), (12288, 34, 12288, torch.float32, 0, 1.0, 5),
              (131072, 34, 5000, torch.bfloat16, 0, 1.0, 6), (262144, 7, 300, torch.float32, 3, 0.25, 7), (262144, 66, 52428, torch.float16, 0, 1.0, 8)]
    for rows, features, p, dtype, pad, scale, seed in cases:
        g = torch.Generator().manual_seed(seed)
        wide = torch.randn(rows, features + pad, generator=g).to(dtype).to(DEV)
        x = wide[:, :features]
        idx = torch.randint(0, rows, (p, ), generator=g).to(DEV)
        if seed == 1:
            idx = (idx // 16) * 16                                           # every sample in ONE residue class of a 16 x 16 split: list overflow
        got = cabi.sampled_dct(x, idx, scale)
        want = fewbit.fft.dct(x.double(), dim=0, norm='ortho')[idx] * scale
        err = (got.double() - want).abs()
        tol = REL[dtype] * want.abs() + 3e-6 * float(want.abs().max()) + 1e-30
        assert bool((err <= tol).all()), (rows, features, p, dtype, pad, float((err / tol).max()))
        assert not torch.isnan(wide).any()
        # the rows as a function of a seed: the same bits as the explicit call on those rows
        of_seed = cabi.sampled_rows(seed, rows, p).to(DEV)
        assert torch.equal(cabi.sampled_dct_seeded(x, p, seed, scale), cabi.sampled_dct(x, of_seed, scale)), (rows, features, p, dtype, pad)


def test_the_seeded_dct_estimator_has_the_mean_and_spread_of_the_formulation_with_drawn_rows():
    """The rows of a seed are another stream than the reference's T.multinomial draw, so this link is statistical (like the dense sketches'
    in tests/test_gpu_linear.py): over 3000 draws each, linear_grp(matmul='dct') on the kernel pair (rows of a seed) and on the torch.fft
    formulation (rows from randint -- the path tests/test_gpu_linear.py pins to the reference's outputs) both average to the exact weight
    gradient, and their mean squared deviations from it agree within 5 % (standard error of the ratio ~ 2 %).  256 and 768 rows."""
    from fewbit_amd import linear
    for rows in (256, 768):
        g = torch.Generator().manual_seed(rows)
        x = torch.randn(rows, 12, generator=g).to(DEV)
        w = (torch.randn(6, 12, generator=g) * 0.3).to(DEV)
        gy = torch.randn(rows, 6, generator=g).to(DEV)
        exact = gy.T @ x
        p, draws = rows // 4, 3000
        torch.manual_seed(9)
        msd = {}
        for native in (True, False):
            prev = linear.use_native_sketch(native)
            try:
                acc, dev2 = torch.zeros_like(exact, dtype=torch.float64), torch.zeros((), device=DEV, dtype=torch.float64)
                for _ in range(draws):
                    wi = w.clone().requires_grad_()
                    fewbit.functional.linear_grp(x, wi, None, proj_dim=p, matmul='dct').backward(gy)
                    acc += wi.grad
                    dev2 += ((wi.grad - exact) ** 2).sum()
                msd[native] = float(dev2) / draws
                bias = float(torch.linalg.norm(acc / draws - exact) / torch.linalg.norm(exact))
                spread = (msd[native] / draws) ** 0.5 / float(torch.linalg.norm(exact))     # one sigma of that norm
                assert bias <= 4.0 * spread + 1e-3, (rows, native, bias, spread)
            finally:
                linear.use_native_sketch(prev)
        print(f'\nseeded DCT estimator, {rows} rows: mean squared deviation kernel pair / torch formulation = {msd[True] / msd[False]:.4f}')
        assert abs(msd[True] / msd[False] - 1.0) <= 0.05, (rows, msd)
