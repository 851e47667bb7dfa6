"""fewbit_hip_sketch on the GPU (fewbit_amd/csrc/fewbit_sketch.hip): the random matrix the kernel generates IN REGISTERS is
exactly the host model's (tests/sketch_reference.py evaluates the same Philox stream), and the product equals S . M."""
import pytest
import torch

import sketch_reference as ref
from fewbit_amd import cabi
from helpers import ulp_distance

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(autouse=True)
def native_sketch_on():
    """these tests are about the gfx950 sketch kernel: select it whatever FEWBIT_SKETCH_NATIVE says in the environment"""
    from fewbit_amd import linear
    prev = linear.use_native_sketch(True)
    yield
    linear.use_native_sketch(prev)


@pytest.mark.parametrize('seed', (0, 1, 0x1234567890abcdef, 2**64 - 1))
def test_rademacher_matrix_is_the_models_bit_for_bit(seed):
    for (nr, nc, r0, c0) in ((70, 1000, 0, 0), (33, 515, 1000, 250), (4, 64, 2**31, 2**33 + 8)):
        got = cabi.sketch_matrix('rademacher', torch.bfloat16, seed, nr, nc, r0, c0).cpu()
        assert torch.equal(got, ref.rademacher(seed, nr, nc, r0, c0)), (seed, nr, nc)


@pytest.mark.parametrize('dtype', (torch.bfloat16, torch.float16))
def test_gaussian_matrix_is_the_models_up_to_one_step_of_the_operand_dtype(dtype):
    """same Philox words, same Box-Muller; v_log/v_sqrt/v_sin/v_cos against libm: the rounded operand may differ by one step"""
    for seed, (nr, nc, r0, c0) in ((5, (64, 1024, 0, 0)), (99, (17, 333, 123, 77))):
        got = cabi.sketch_matrix('gaussian', dtype, seed, nr, nc, r0, c0).cpu()
        want = ref.gaussian(seed, nr, nc, dtype, r0, c0)
        d = ulp_distance(got.to(dtype), want.to(dtype))
        # (near zero one operand step is tiny and cos / sin of a quarter turn is an exact 0 in hardware, 6e-17 in libm:
        # there the comparison is absolute)
        ok = (d <= 1) | ((got - want).abs() <= 2.0**-12)
        assert bool(ok.all()), (int(d[~ok].max()), float((got - want).abs().max()))
        assert float((d == 0).float().mean()) > 0.97
        exact = ref.gaussian(seed, nr, nc, dtype, r0, c0, rounded=False)
        assert float((got.double() - exact).abs().max()) < 0.02


def test_moments_of_the_matrix_the_device_generates():
    """2048 x 8192 = 1.7e7 entries of S straight from the device generator (fewbit_hip_sketch_matrix): Gaussian mean 0, variance
    1, kurtosis 3.0 +- 0.02 (sampling error of the kurtosis at this size: 0.0012), no correlation between neighbours in a row,
    between the two streams of a block, between consecutive steps, octet parities, blocks and rows -- in the values and in their
    squares; Rademacher: +-1 in balance.  Unrounded statistics are the host model's (tests/test_sketch.py); these are the operands
    the matrix pipe gets, rounded to bf16 / fp16."""
    nr, nc = 2048, 8192
    for dtype in (torch.bfloat16, torch.float16):
        G = cabi.sketch_matrix('gaussian', dtype, 0xfeedface, nr, nc).double()
        n = G.numel()
        mean, var = float(G.mean()), float(G.var())
        kurt = float(((G - mean)**4).mean()) / var**2
        assert abs(mean) < 1.5e-3 and abs(var - 1.0) < 2.5e-3 and abs(kurt - 3.0) < 0.02, (dtype, mean, var, kurt)
        assert abs(float((G**3).mean())) < 0.01 and abs(float((G**6).mean()) - 15.0) < 0.4
        assert float(G.abs().max()) < 4.9 and float((G.abs() > 3.0).double().mean()) == pytest.approx(0.0027, abs=2e-4)
        for a, b in ((G[:, :-1], G[:, 1:]), (G[:, 0::8], G[:, 4::8]), (G[:, :-16], G[:, 16:]), (G[:, :-8], G[:, 8:]), (G[:, :-256], G[:, 256:]),
                     (G[:-1], G[1:]), (G[:-32], G[32:])):
            assert abs(float((a * b).mean())) < 2e-3 and abs(float((a * a * b * b).mean()) - 1.0) < 6e-3, dtype
    R = cabi.sketch_matrix('rademacher', torch.bfloat16, 0xfeedface, nr, nc).double()
    assert set(R.unique().tolist()) == {-1.0, 1.0} and abs(float(R.mean())) < 1e-3
    assert float(R.sum(1).abs().max()) < 6 * nc**0.5 and float(R.sum(0).abs().max()) < 6 * nr**0.5          # every row and column in balance
    for a, b in ((R[:, :-1], R[:, 1:]), (R[:, :-16], R[:, 16:]), (R[:, :-256], R[:, 256:]), (R[:-1], R[1:])):
        assert abs(float((a * b).mean())) < 1.5e-3


_MODEL_MATRICES = {}


def _model_matrix(dist, seed, proj, rows, dtype):
    """the host model's S for these arguments (generated once per module run: the tune sweeps below ask for the same matrix several times)"""
    key = (dist, seed, proj, rows, dtype)
    if key not in _MODEL_MATRICES:
        if len(_MODEL_MATRICES) >= 4:                                  # (bounded: a 1290 x 70000 matrix in float64 is 0.7 GB)
            _MODEL_MATRICES.pop(next(iter(_MODEL_MATRICES)))
        _MODEL_MATRICES[key] = ref.matrix(dist, seed, proj, rows, dtype).double()
    return _MODEL_MATRICES[key]


def _product_case(dist, dtype, rows, features, proj, seed, ld=None, scale=1.0):
    g = torch.Generator().manual_seed(rows * 31 + features)
    m = torch.randn(rows, ld or features, generator=g).to(dtype)[:, :features]
    got = cabi.sketch(dist, m.to(DEV) if ld is None else m.to(DEV), proj, seed, scale)
    assert got.shape == (proj, features) and got.dtype == dtype
    S = _model_matrix(dist, seed, proj, rows, dtype)
    op = torch.float16 if dtype == torch.float16 else torch.bfloat16
    mm = m.to(op).double()
    want = scale * (S @ mm)
    bound = abs(scale) * (S.abs() @ mm.abs())                      # scale of the individual sums
    err = (got.cpu().double() - want).abs()
    # fp32 accumulation of exact products (+ for Gaussian one operand step on a few entries), then one rounding to `dtype`
    out_eps = {torch.float32: 2.0**-22, torch.float16: 2.0**-10, torch.bfloat16: 2.0**-7}[dtype]
    slack = (2.0**-8 if dist == 'gaussian' else 0.0) * bound / max(rows, 1)**0.5 * 8
    if dist == 'gaussian':      # near zero an operand step is tiny and cos / sin of a quarter turn is an exact 0 in hardware, 6e-17 in libm
        slack = slack + abs(scale) * 2.0**-12 * mm.abs().sum(0, keepdim=True)      # (the matrix test's absolute 2^-12 per element of S)
    plan = cabi.describe_sketch(dist, rows, features, proj, dtype)
    if plan['partial_sums'] == 'bf16':           # every slice's sum makes its way to the reduce kernel rounded to bf16: at most 2^-8 of it each
        assert plan['grid'][2] > 1 and (dtype == torch.bfloat16 or (dtype == torch.float32 and plan['converted_to_bf16_first']))
        ks = plan['k_slice']
        slack = slack + abs(scale) * 2.0**-8 * sum((S[:, z:z + ks] @ mm[z:z + ks]).abs() for z in range(0, rows, ks)) * 1.01
    floor = 2.0**-24 if dtype == torch.float16 else 1e-30          # fp16 results below 6e-5 are subnormal: steps of 2^-24
    assert bool((err <= out_eps * want.abs() + 1e-5 * bound + slack + floor).all()), (dist, dtype, rows, features, proj, float((err / (bound + 1e-30)).max()))
    return got


@pytest.mark.parametrize('path', ('fused', 'fused_halves', 'memory'))
@pytest.mark.parametrize('dist', ('rademacher', 'gaussian'))
@pytest.mark.parametrize('dtype', (torch.float32, torch.bfloat16, torch.float16))
def test_product_equals_the_model_matrix_times_m(dist, dtype, path):
    """`fused`: S generated inside the product kernel; `fused_halves`: the same on the 128 x 512 tile, whose wave pairs hand each
    other their A fragments through LDS; `memory`: (Gaussian) S written to the workspace once as MFMA fragments and read back by
    the product kernel -- the same S, the same products"""
    cabi.tune_sketch_halves(2 if path == 'fused_halves' else 1)
    cabi.tune_sketch_materialise(1 if path == 'memory' else 0)
    try:
        for rows, features, proj in ((64, 256, 128), (100, 37, 5), (1000, 264, 130), (257, 8, 1), (4096, 512, 256), (3000, 770, 200), (2048, 1024, 300)):
            plan = cabi.describe_sketch(dist, rows, features, proj, dtype)
            # ("whenever possible" keeps the rules that make the path worth taking: more than one 256-feature column tile to share S with)
            from_memory = path == 'memory' and dist == 'gaussian' and features > 256 and (dtype != torch.float32 or plan['converted_to_bf16_first'])
            assert (plan['s_fragment_bytes'] > 0) == from_memory == ('from memory' in plan['kernel']), plan
            if from_memory:
                assert plan['s_fragment_bytes'] == (-(-proj // 256) * 8 * -(-rows // 256) * 16 + 4) * 1024 and plan['workspace_bytes'] >= plan['s_fragment_bytes']
            _product_case(dist, dtype, rows, features, proj, seed=rows + 17)
        _product_case(dist, dtype, 512, 100, 64, seed=3, ld=136, scale=0.125)         # a strided view, a scale
    finally:
        cabi.tune_sketch_halves(-1)
        cabi.tune_sketch_materialise(-1)


def test_gaussian_fragments_from_memory_are_the_fused_kernels_matrix():
    """the policy (Gaussian, 16-bit operand, more than one column tile): S once into the workspace, product kernel reads it -- bit
    for bit the fused kernel's result when both slice the rows alike, deterministic, also for fp32 input that is rounded to bf16
    first and for a device-resident seed"""
    assert 'from memory' in cabi.describe_sketch('gaussian', 16384, 3072, 3276)['kernel']
    assert 'from memory' in cabi.describe_sketch('gaussian', 16384, 768, 3276)['kernel']
    assert 'from memory' not in cabi.describe_sketch('gaussian', 16384, 768, 3276, torch.float32)['kernel']        # fp32 input: the fused kernel by policy ...
    assert 'from memory' in cabi.describe_sketch('gaussian', 16384, 3072, 3276, torch.float32)['kernel']           # ... on layers narrower than 2048 features
    try:
        cabi.tune_sketch_materialise(1)                                                                             # ... unless asked for
        assert 'from memory' in cabi.describe_sketch('gaussian', 16384, 768, 3276, torch.float32)['kernel']        # (converted to bf16 first)
        assert 'from memory' not in cabi.describe_sketch('gaussian', 16384, 768, 1000, torch.float32)['kernel']    # (fp32 operand staged in the kernel)
    finally:
        cabi.tune_sketch_materialise(-1)
    assert 'from memory' not in cabi.describe_sketch('gaussian', 16384, 256, 3276)['kernel']                       # one column tile: nothing to share
    assert 'from memory' not in cabi.describe_sketch('rademacher', 16384, 3072, 3276)['kernel']
    assert cabi.describe_sketch('gaussian', 2**21, 768, 2**10)['s_fragment_bytes'] == 0                             # 4 GiB of fragments: the fused kernel
    g = torch.Generator().manual_seed(9)
    try:
        for dtype, rows, features, proj, z in ((torch.bfloat16, 5000, 776, 300, 2), (torch.float16, 2049, 520, 257, 1), (torch.float32, 3000, 384, 1400, 3)):
            m = torch.randn(rows, features, generator=g).to(dtype).to(DEV)
            cabi.tune_sketch_slices(z)
            cabi.tune_sketch_halves(1)
            got = {}
            for mat in (0, 1):
                cabi.tune_sketch_materialise(mat)
                cabi.tune_sketch_waves(8)                    # the same tile and slicing for both: the same fp32 sums in the same order
                assert ('from memory' in cabi.describe_sketch('gaussian', rows, features, proj, dtype)['kernel']) == bool(mat)
                got[mat] = cabi.sketch('gaussian', m, proj, 77, 0.5)
                assert torch.equal(got[mat], cabi.sketch('gaussian', m, proj, 77, 0.5))
            assert torch.equal(got[0], got[1]), (dtype, float((got[0].float() - got[1].float()).abs().max()))
            cabi.tune_sketch_waves(4)
            assert torch.allclose(cabi.sketch('gaussian', m, proj, 77, 0.5).float(), got[1].float(), rtol=2e-2, atol=0.5)
            word = torch.tensor([77], dtype=torch.int64, device=DEV)
            cabi.tune_sketch_waves(8)
            assert torch.equal(cabi.sketch('gaussian', m, proj, word, 0.5), got[1])
    finally:
        cabi.tune_sketch_slices(-1)
        cabi.tune_sketch_halves(-1)
        cabi.tune_sketch_waves(-1)
        cabi.tune_sketch_materialise(-1)


def test_row_slices_are_deterministic_and_agree():
    """split K: every slicing gives the same sums up to fp32 re-association, and a repeated call the same BITS"""
    m = torch.randn(8192, 384, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).to(DEV)
    try:
        outs = {}
        for z in (1, 2, 4, 8):
            cabi.tune_sketch_slices(z)
            assert cabi.describe_sketch('rademacher', 8192, 384, 200)['grid'][2] == z
            a = cabi.sketch('rademacher', m.float(), 200, 42)
            b = cabi.sketch('rademacher', m.float(), 200, 42)
            assert torch.equal(a, b)
            outs[z] = a
        for z in (2, 4, 8):
            assert torch.allclose(outs[z], outs[1], rtol=1e-5, atol=1e-3)
        cabi.tune_sketch_slices(-1)
        cabi.tune_sketch_halves(1)
        for w in (4, 8):                                 # both tile heights: the same sums
            cabi.tune_sketch_waves(w)
            assert cabi.describe_sketch('rademacher', 8192, 384, 200)['threads'] == 64 * w
            for dist in ('rademacher', 'gaussian'):
                outs[(dist, w)] = cabi.sketch(dist, m, 200, 42)
        cabi.tune_sketch_waves(-1)
        cabi.tune_sketch_halves(2)                       # the 128 x 512 tile (A fragments shared through LDS), both sketches
        cabi.tune_sketch_materialise(0)
        for dist in ('rademacher', 'gaussian'):
            assert '128x512' in cabi.describe_sketch(dist, 8192, 384, 200)['kernel']
            outs[(dist, 'wide')] = cabi.sketch(dist, m, 200, 42)
        for dist in ('rademacher', 'gaussian'):
            assert torch.allclose(outs[(dist, 4)].float(), outs[(dist, 8)].float(), rtol=2e-2, atol=0.5)
            assert torch.allclose(outs[(dist, 4)].float(), outs[(dist, 'wide')].float(), rtol=2e-2, atol=0.5)
    finally:
        cabi.tune_sketch_slices(-1)
        cabi.tune_sketch_waves(-1)
        cabi.tune_sketch_halves(-1)
        cabi.tune_sketch_materialise(-1)
    plan = cabi.describe_sketch('rademacher', 16384, 3072, 1638)
    assert plan['threads'] in (256, 512) and plan['grid'][0] == 12 and plan['grid'][1] == -(-1638 // (plan['threads'] // 2)) and plan['grid'][2] >= 1
    plan = cabi.describe_sketch('gaussian', 16384, 3072, 1638)                       # the policy: S from memory, the Rademacher plan's tile
    assert 'from memory' in plan['kernel'] and plan['grid'][0] == 12 and plan['s_fragment_bytes'] == (7 * 8 * 64 * 16 + 4) * 1024
    try:
        cabi.tune_sketch_materialise(0)                                              # the fused kernel: the 128 x 512 tile for wide 16-bit layers
        plan = cabi.describe_sketch('gaussian', 16384, 3072, 1638)
        assert '128x512' in plan['kernel'] and plan['grid'][:2] == [6, 13] and plan['lds_bytes'] == 163840 and plan['s_fragment_bytes'] == 0
    finally:
        cabi.tune_sketch_materialise(-1)


def test_fp32_input_rounded_first_with_and_without_gaussian_fragments_from_memory():
    """fp32 input with many row tiles: conversion pass, then the fused kernel (the policy for fp32 input narrower than 2048 features) or
    -- forced, or by policy on a wide layer -- fragment pass + product from memory; ragged widths, a row stride, a very long and a
    very short input"""
    try:
        for rows, features, proj, ld in ((3000, 770, 1400, None), (520, 264, 1300, 272), (70000, 40, 1290, None), (300, 1032, 2000, 1040), (700, 2056, 1300, None)):
            for mem in (-1, 1, 0):                   # the policy keeps narrow fp32 input on the fused kernel; 1 lifts that width rule; 0 never
                cabi.tune_sketch_materialise(mem)
                plan = cabi.describe_sketch('gaussian', rows, features, proj, torch.float32)
                want = features > 256 and (mem == 1 or (mem == -1 and features >= 2048))
                assert plan['converted_to_bf16_first'] is True and ('from memory' in plan['kernel']) == want, (plan, mem)
                _product_case('gaussian', torch.float32, rows, features, proj, seed=rows, ld=ld)
    finally:
        cabi.tune_sketch_materialise(-1)


def test_bf16_partial_sums_of_sliced_bf16_products():
    """bf16 result + sliced rows: the slices' sums cross the workspace in bf16 (half the bytes), are added in fp32 in slice order,
    deterministic; against fp32 partial sums the result moves by at most the roundings of the slices' sums"""
    m = torch.randn(8192, 392, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).to(DEV)
    try:
        for dist in ('rademacher', 'gaussian'):
            for z in (2, 4, 7):
                cabi.tune_sketch_slices(z)
                got = {}
                for p16 in (0, 1):
                    cabi.tune_sketch_partials(p16)
                    plan = cabi.describe_sketch(dist, 8192, 392, 200)
                    assert plan['partial_sums'] == ('bf16' if p16 else 'fp32') and plan['grid'][2] == z
                    partial = z * 200 * 392 * (2 if p16 else 4)                 # then (Gaussian, wider than a tile) S's fragments, 256-aligned
                    want_ws = partial if plan['s_fragment_bytes'] == 0 else -(-partial // 256) * 256 + plan['s_fragment_bytes']
                    assert plan['workspace_bytes'] == want_ws == cabi.sketch_workspace_bytes(dist, 8192, 392, 200)
                    got[p16] = cabi.sketch(dist, m, 200, 11, 0.5)
                    assert torch.equal(got[p16], cabi.sketch(dist, m, 200, 11, 0.5))
                ks = plan['k_slice']
                S = cabi.sketch_matrix(dist, torch.bfloat16, 11, 200, 8192).double()
                parts = sum((S[:, k:k + ks] @ m[k:k + ks].double()).abs() for k in range(0, 8192, ks))
                want = 0.5 * (S @ m.double())
                err = (got[1].double() - want).abs()
                assert bool((err <= 2.0**-8 * want.abs() + 0.5 * 2.0**-8 * parts * 1.01 + 1e-3).all()), (dist, z, float(err.max()))
                assert not torch.equal(got[0], got[1]) and float((got[0].float() - got[1].float()).abs().max()) < 0.5 * float(parts.max()) * 2.0**-7
            # a ragged feature count takes the element-wise reduce kernel, an fp16 / fp32 result keeps fp32 partial sums
            cabi.tune_sketch_partials(-1)
            cabi.tune_sketch_slices(3)
            _product_case(dist, torch.bfloat16, 3000, 389, 130, seed=4)
            assert cabi.describe_sketch(dist, 8192, 392, 200, torch.float16)['partial_sums'] == 'fp32'
            assert cabi.describe_sketch(dist, 8192, 392, 200, torch.float32)['partial_sums'] == 'fp32'           # (fp32 operand staged in the kernel: p <= 1280)
            assert cabi.describe_sketch(dist, 8192, 392, 1400, torch.float32)['partial_sums'] == 'bf16'          # (rounded to bf16 first: bf16 operands)
        cabi.tune_sketch_slices(1)
        assert cabi.describe_sketch('rademacher', 8192, 392, 200)['partial_sums'] is None
    finally:
        cabi.tune_sketch_slices(-1)
        cabi.tune_sketch_partials(-1)


def test_empty_and_degenerate_shapes():
    assert cabi.sketch('gaussian', torch.zeros(0, 16, device=DEV), 4, 1).abs().sum() == 0
    assert cabi.sketch('rademacher', torch.zeros(16, 0, device=DEV), 4, 1).shape == (4, 0)
    assert cabi.sketch('rademacher', torch.ones(16, 8, device=DEV), 0, 1).shape == (0, 8)
    with pytest.raises(cabi.FewbitHipError):
        cabi.sketch('rademacher', torch.ones(16, 8), 4, 1)                          # host tensor
    with pytest.raises(cabi.FewbitHipError):
        cabi.sketch('rademacher', torch.ones(8, 16, device=DEV).t(), 4, 1)          # features not contiguous


@pytest.mark.parametrize('dist', ('rademacher', 'gaussian'))
@pytest.mark.parametrize('dtype,rows,features,ld', ((torch.bfloat16, 2**21 + 5, 1032, 1040), (torch.float32, 2**20 + 77, 1100, 1100)))
def test_inputs_beyond_four_gib_against_the_materialised_matrix(dist, dtype, rows, features, ld):
    """Byte offsets past 2^32 (4.4 GB bf16 with a row stride, 4.6 GB fp32): the product against S itself (fewbit_hip_sketch_matrix,
    the same entries rounded the same way) times m in fp64, in row chunks.  fp32 accumulation over 2^21 rows: the error of a
    sum is a random walk of roundings, bounded here by 2^-24 x sqrt(rows) x 16 relative to sum |s||m|."""
    es = torch.empty(0, dtype=dtype).element_size()
    assert rows * ld * es > 2**32
    proj, seed = 48, 0xfeedbeefcafe
    g = torch.Generator(device=DEV).manual_seed(5)
    m = torch.randn(rows, ld, generator=g, device=DEV, dtype=torch.float32).to(dtype)[:, :features]
    got = cabi.sketch(dist, m, proj, seed)
    op = torch.float16 if dtype == torch.float16 else torch.bfloat16
    want = torch.zeros(proj, features, dtype=torch.float64, device=DEV)
    bound = torch.zeros_like(want)
    step = 2**18
    for r0 in range(0, rows, step):
        n = min(step, rows - r0)
        S = cabi.sketch_matrix(dist, dtype, seed, proj, n, 0, r0).double()
        mm = m[r0:r0 + n].to(op).double()
        want += S @ mm
        bound += S.abs() @ mm.abs()
    err = (got.double() - want).abs()
    out_eps = 2.0**-22 if dtype == torch.float32 else 2.0**-7
    assert bool((err <= out_eps * want.abs() + 2.0**-24 * rows**0.5 * 16 * bound).all()), float((err / bound).max())
    assert float(want.abs().mean()) > 100.0                                 # (a sum over 2^21 rows, not a field of zeros)
    # the far end of the buffer matters: zero the last rows and the result moves by exactly their contribution
    tail = 300
    m2 = m.clone()
    m2[rows - tail:] = 0
    S = cabi.sketch_matrix(dist, dtype, seed, proj, tail, 0, rows - tail).double()
    moved = (got.double() - cabi.sketch(dist, m2, proj, seed).double())
    assert torch.allclose(moved, S @ m[rows - tail:].to(op).double(), rtol=0, atol=float(want.abs().max()) * 2 * out_eps + 1e-2)


def test_half_a_gigabyte_of_fragments_against_the_materialised_matrix():
    """S from memory with a long input: 2^20 + 3 rows, 200 rows of S = 537 MB of fragments (fragment offsets past 2^29, 4097 blocks of
    256 rows, the last one nearly empty), sliced rows with bf16 partial sums -- against S itself times m in fp64, in row chunks"""
    rows, features, proj, seed = 2**20 + 3, 264, 200, 0x5eed0123456789
    plan = cabi.describe_sketch('gaussian', rows, features, proj)
    assert 'from memory' in plan['kernel'] and plan['s_fragment_bytes'] == (8 * 4097 * 16 + 4) * 1024 > 2**29 and plan['partial_sums'] == 'bf16'
    m = torch.randn(rows, features, generator=torch.Generator(device=DEV).manual_seed(6), device=DEV, dtype=torch.float32).to(torch.bfloat16)
    got = cabi.sketch('gaussian', m, proj, seed)
    assert torch.equal(got, cabi.sketch('gaussian', m, proj, seed))
    want = torch.zeros(proj, features, dtype=torch.float64, device=DEV)
    bound = torch.zeros_like(want)
    parts = torch.zeros_like(want)
    ks, step = plan['k_slice'], 2**17
    for z0 in range(0, rows, ks):
        pz = torch.zeros_like(want)
        for r0 in range(z0, min(z0 + ks, rows), step):
            n = min(step, min(z0 + ks, rows) - r0)
            S = cabi.sketch_matrix('gaussian', torch.bfloat16, seed, proj, n, 0, r0).double()
            mm = m[r0:r0 + n].double()
            pz += S @ mm
            bound += S.abs() @ mm.abs()
        want += pz
        parts += pz.abs()
    err = (got.double() - want).abs()
    assert bool((err <= 2.0**-7 * want.abs() + 2.0**-8 * parts * 1.01 + 2.0**-24 * rows**0.5 * 16 * bound).all()), float((err / bound).max())
    assert float(want.abs().mean()) > 100.0
    try:                                             # the fused kernel on the same input: the same sums up to the slices' roundings
        cabi.tune_sketch_materialise(0)
        fused = cabi.sketch('gaussian', m, proj, seed)
    finally:
        cabi.tune_sketch_materialise(-1)
    assert bool(((fused.double() - got.double()).abs() <= 2.0**-6 * want.abs() + 2.0**-7 * parts).all())


def test_unsupported_sizes_are_refused_not_wrapped():
    wide = torch.zeros(8, 8 * 2**20, dtype=torch.bfloat16, device=DEV)      # a K stage of this row stride spans > 2 GiB
    with pytest.raises(cabi.FewbitHipError, match='leading dimension'):
        cabi.sketch('rademacher', wide[:, :64], 4, 1)


def test_estimator_is_unbiased_on_the_gpu_kernel():
    """E[(S G)^T (S X)] / proj = G^T X: mean over seeds of the kernel's own products converges to the exact product"""
    g = torch.Generator().manual_seed(0)
    x, gy = torch.randn(512, 48, generator=g).to(DEV), torch.randn(512, 40, generator=g).to(DEV)
    exact = gy.T @ x
    for dist in ('rademacher', 'gaussian'):
        acc = torch.zeros_like(exact)
        n = 300
        for seed in range(n):
            acc += cabi.sketch(dist, gy, 64, seed).T @ cabi.sketch(dist, x, 64, seed, 1.0 / 64)
        rel = float(torch.linalg.norm(acc / n - exact) / torch.linalg.norm(exact))
        assert rel < 0.25, (dist, rel)                                    # one draw: ~sqrt(512/64) = 2.8; mean of 300: ~0.16


def test_seeded_fuzz_of_shapes_dtypes_strides_and_tiles():
    """60 random cases (FEWBIT_SKETCH_FUZZ_CASES=N widens the sweep: a one-off soak, profiles/r04_sketch_soak_fuzz.txt) around the tile edges (128 / 256 rows of S, 256 / 512 features, stages of 64 / 128 rows, 256-row
    Rademacher blocks, 1024-row slices): distribution, dtype, ragged sizes, a row stride, a scale, the tile height / width and
    the slicing are drawn per case; every product must equal the host model's"""
    import random
    rnd = random.Random(20260402)
    rnd_convert = random.Random(7)                       # (its own stream: the cases above keep their sequence)
    rnd_partials = random.Random(8)
    rnd_memory = random.Random(9)
    edges_r = (1, 7, 8, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 2047, 2048, 3000)
    edges_f = (1, 8, 9, 40, 255, 256, 257, 264, 511, 512, 520, 768, 1032)
    edges_p = (1, 31, 32, 33, 127, 128, 129, 255, 256, 257, 300)
    try:
        import os
        for case in range(int(os.environ.get('FEWBIT_SKETCH_FUZZ_CASES', '60'))):
            dist = rnd.choice(('rademacher', 'gaussian'))
            dtype = rnd.choice((torch.float32, torch.bfloat16, torch.float16))
            rows, features, proj = rnd.choice(edges_r), rnd.choice(edges_f), rnd.choice(edges_p)
            cabi.tune_sketch_waves(rnd.choice((-1, 4, 8)))
            cabi.tune_sketch_halves(rnd.choice((-1, 1, 2)))
            cabi.tune_sketch_slices(rnd.choice((-1, 1, 2, 3)))
            cabi.tune_sketch_convert(rnd_convert.choice((-1, 0, 1)))         # fp32 input: one conversion pass first, or not
            cabi.tune_sketch_partials(rnd_partials.choice((-1, 0, 1)))       # bf16 result, sliced rows: bf16 or fp32 partial sums
            cabi.tune_sketch_materialise(rnd_memory.choice((-1, 0, 1)))      # Gaussian: S from memory, or generated in the product kernel
            ld = features + rnd.choice((0, 0, 8, 3)) if features > 1 else None
            try:
                _product_case(dist, dtype, rows, features, proj, seed=rnd.getrandbits(64), ld=ld if ld != features else None,
                              scale=rnd.choice((1.0, 1.0 / proj, -0.5)))
            except AssertionError as e:
                raise AssertionError(f'case {case}: {dist} {dtype} {rows} x {features} (ld {ld}) proj {proj}, plan {cabi.describe_sketch(dist, rows, features, proj, dtype)}') from e
    finally:
        cabi.tune_sketch_waves(-1)
        cabi.tune_sketch_halves(-1)
        cabi.tune_sketch_slices(-1)
        cabi.tune_sketch_convert(-1)
        cabi.tune_sketch_partials(-1)
        cabi.tune_sketch_materialise(-1)


def test_fp32_input_converted_to_bf16_first_gives_the_same_products():
    """p > 1280 (six row tiles re-reading M): fp32 input is rounded to bf16 once, into the workspace, and the bf16-input kernel
    runs on the copy; the result is still fp32 from the fp32 sums.  Same rounding of M either way -> the same products, up to
    the association of the fp32 sums where the two paths slice the rows differently."""
    assert cabi.describe_sketch('rademacher', 16384, 768, 3276, torch.float32)['converted_to_bf16_first'] is True
    assert cabi.describe_sketch('gaussian', 16384, 768, 1280, torch.float32)['converted_to_bf16_first'] is False
    assert cabi.describe_sketch('rademacher', 16384, 768, 3276, torch.bfloat16)['converted_to_bf16_first'] is False
    try:
        cabi.tune_sketch_partials(2)           # (fp32 partial sums for the fp32 result: the two routes are then the same sums up to association)
        for dist in ('rademacher', 'gaussian'):
            for rows, features, proj, ld in ((3000, 770, 200, None), (4096, 512, 1400, None), (512, 100, 64, 136), (2048, 1024, 1300, 1032), (1000, 37, 5, None)):
                g = torch.Generator().manual_seed(rows)
                m = torch.randn(rows, ld or features, generator=g).to(DEV)[:, :features]
                got = {}
                for convert in (0, 1):
                    cabi.tune_sketch_convert(convert)
                    plan = cabi.describe_sketch(dist, rows, features, proj, torch.float32)
                    assert plan['converted_to_bf16_first'] is bool(convert)
                    assert plan['workspace_bytes'] == cabi.sketch_workspace_bytes(dist, rows, features, proj, torch.float32)
                    if convert:
                        assert plan['workspace_bytes'] >= rows * features * 2 + proj * features * 4
                    got[convert] = cabi.sketch(dist, m, proj, 99, 0.25)
                    assert got[convert].dtype == torch.float32
                    assert torch.equal(got[convert], cabi.sketch(dist, m, proj, 99, 0.25))           # deterministic
                bound = 0.25 * (cabi.sketch_matrix(dist, torch.float32, 99, proj, rows).abs() @ m.abs())
                assert bool(((got[0] - got[1]).abs() <= 2.0**-20 * bound + 1e-30).all()), (dist, rows, features, proj)
                _product_case(dist, torch.float32, rows, features, proj, seed=5, ld=ld)                # (convert = 1 still set)
                cabi.tune_sketch_partials(-1)      # the policy: sliced rows of a converted input exchange bf16 partial sums, fp32 result
                cabi.tune_sketch_slices(3)
                plan = cabi.describe_sketch(dist, rows, features, proj, torch.float32)
                assert plan['partial_sums'] == ('bf16' if plan['grid'][2] > 1 else 'fp32') and (plan['grid'][2] > 1) == (rows >= 2048)
                _product_case(dist, torch.float32, rows, features, proj, seed=6, ld=ld)
                cabi.tune_sketch_slices(-1)
                cabi.tune_sketch_partials(2)
    finally:
        cabi.tune_sketch_convert(-1)
        cabi.tune_sketch_partials(-1)
        cabi.tune_sketch_slices(-1)


def test_sketch_and_randomized_layer_capture_into_a_hip_graph(monkeypatch):
    """no call of the path synchronises or reads back from the device: the kernel (and a whole randomized layer step with its seed
    drawn on the host) can be captured once and replayed on new data"""
    import fewbit
    m = torch.zeros(2048, 512, device=DEV, dtype=torch.bfloat16)
    out = torch.empty(130, 512, device=DEV, dtype=torch.bfloat16)
    ws = torch.empty(max(cabi.sketch_workspace_bytes('rademacher', 2048, 512, 130), 1), dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        cabi.sketch('rademacher', m, 130, 5, 1.0, out=out, workspace=ws)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cabi.sketch('rademacher', m, 130, 5, 1.0, out=out, workspace=ws)
    data = torch.randn(2048, 512, generator=torch.Generator().manual_seed(2)).to(torch.bfloat16)
    m.copy_(data.to(DEV))
    g.replay()
    torch.cuda.synchronize()
    want = ref.rademacher(5, 130, 2048).double() @ data.double()
    assert float((out.cpu().double() - want).abs().max() / want.abs().max()) < 2.0**-7
    # the Gaussian sketch with S from memory is three launches (fragments, product, reduce) on the stream: captured and replayed, seed
    # by value and seed in device memory, it gives the eager call's bits
    assert 'from memory' in cabi.describe_sketch('gaussian', 2048, 512, 130)['kernel']
    gws = torch.empty(cabi.sketch_workspace_bytes('gaussian', 2048, 512, 130), dtype=torch.uint8, device=DEV)
    gout = torch.empty(130, 512, device=DEV, dtype=torch.bfloat16)
    word = torch.tensor([77], dtype=torch.int64, device=DEV)
    for seed in (77, word):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            cabi.sketch('gaussian', m, 130, seed, 0.5, out=gout, workspace=gws)
        torch.cuda.current_stream().wait_stream(side)
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg):
            cabi.sketch('gaussian', m, 130, seed, 0.5, out=gout, workspace=gws)
        gout.zero_()
        gg.replay()
        torch.cuda.synchronize()
        assert torch.equal(gout, cabi.sketch('gaussian', m, 130, 77, 0.5)) and float(gout.float().abs().max()) > 0
        del gg
    # a layer step: forward + backward of RandomizedLinear inside one graph
    lin = fewbit.RandomizedLinear(64, 32, proj_dim_ratio=0.25, matmul='rademacher', device=DEV)
    x = torch.zeros(512, 64, device=DEV, requires_grad=True)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            gw, = torch.autograd.grad(lin(x).sum(), lin.weight)
    torch.cuda.current_stream().wait_stream(side)
    from fewbit_amd import linear
    base = 0x5eed5eed
    monkeypatch.setattr(linear, '_draw_seed', lambda generator: base)
    c0 = int(linear._replay_counter(torch.device(DEV)))
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        gw, = torch.autograd.grad(lin(x).sum(), lin.weight)
    with torch.no_grad():
        x.copy_(torch.randn(512, 64, device=DEV))
    g2.replay()
    torch.cuda.synchronize()
    # the replayed step IS the eager estimate for the seed the recorded seed kernel derived (bias and all): p = 128 of 512 rows
    seed = cabi.mix_sketch_seed(base, c0)
    want = cabi.sketch('rademacher', torch.ones(512, 32, device=DEV), 128, seed).T @ cabi.sketch('rademacher', x.detach(), 128, seed, 1.0 / 128)
    assert gw.shape == want.shape and torch.allclose(gw, want, rtol=1e-4, atol=1e-3), float((gw - want).abs().max())
    exact = torch.ones(512, 32, device=DEV).T @ x.detach()
    assert float(torch.linalg.norm(gw - exact) / torch.linalg.norm(exact)) < 4.0          # (and an estimate of the exact product: one draw ~ sqrt(rows / p) = 2)


def _splitmix(base, count):
    M = 2**64 - 1
    x = (base + (count + 1) * 0x9E3779B97F4A7C15) & M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
    return x ^ (x >> 31)


def test_seed_in_device_memory_is_the_same_sketch_as_the_seed_by_value():
    counter = torch.tensor([41], dtype=torch.int64, device=DEV)
    base = 0xfedcba9876543210
    word = cabi.next_sketch_seed(counter, base)
    assert int(counter) == 42
    seed = int(word) & (2**64 - 1)
    assert seed == cabi.mix_sketch_seed(base, 41) == _splitmix(base, 41)
    assert cabi.mix_sketch_seed(0, 0) == _splitmix(0, 0) and cabi.mix_sketch_seed(2**64 - 1, 2**64 - 1) == _splitmix(2**64 - 1, 2**64 - 1)
    for dist, dtype, shape, proj in (('rademacher', torch.bfloat16, (3000, 520), 300), ('gaussian', torch.float16, (1111, 1024), 129),
                                     ('gaussian', torch.float32, (4096, 264), 64), ('rademacher', torch.float32, (9000, 768), 700)):
        m = torch.randn(*shape, device=DEV).to(dtype)
        assert torch.equal(cabi.sketch(dist, m, proj, word, 0.5), cabi.sketch(dist, m, proj, seed, 0.5)), (dist, dtype)
    with pytest.raises(cabi.FewbitHipError):
        cabi.sketch('rademacher', torch.ones(16, 8, device=DEV), 4, torch.zeros(1, dtype=torch.int64))            # a host word
    with pytest.raises(cabi.FewbitHipError):
        cabi.sketch('rademacher', torch.ones(16, 8, device=DEV), 4, torch.zeros(2, dtype=torch.int64, device=DEV))


@pytest.mark.parametrize('kind', ('rademacher', 'gaussian'))
def test_every_replay_of_a_captured_layer_step_draws_a_fresh_sketch(kind, monkeypatch):
    """A seed recorded by value would replay ONE matrix for ever; the recorded seed kernel derives it from (the host draw made
    at capture time, a device counter the replays advance): replay r of the graph equals the eager product with seed
    mix(base, c0 + r), the backward inside the same replay meets the forward's matrix, and the replays differ."""
    import fewbit
    from fewbit_amd import linear
    lin = fewbit.RandomizedLinear(64, 32, proj_dim=96, matmul=kind, bias=False, device=DEV)
    x = torch.randn(512, 64, device=DEV, requires_grad=True)
    wgt = torch.randn(512, 32, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                          # the warm-up every capture needs (creates the replay counter too)
        torch.autograd.grad((lin(x) * wgt).sum(), lin.weight)
    torch.cuda.current_stream().wait_stream(side)
    base = 0x1234567
    monkeypatch.setattr(linear, '_draw_seed', lambda generator: base)
    counter = linear._replay_counter(torch.device(DEV))
    c0 = int(counter)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gw, = torch.autograd.grad((lin(x) * wgt).sum(), lin.weight)
    assert int(counter) == c0                              # recording runs nothing
    seen = []
    for r in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert int(counter) == c0 + r + 1
        seed = cabi.mix_sketch_seed(base, c0 + r)
        want = cabi.sketch(kind, wgt, 96, seed).T @ cabi.sketch(kind, x.detach(), 96, seed, 1.0 / 96)
        assert torch.allclose(gw, want, rtol=1e-4, atol=1e-3), (kind, r, float((gw - want).abs().max()))
        seen.append(gw.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    # outside a capture nothing changes: the seed is a host draw passed by value, the counter stays where it is
    gw2, = torch.autograd.grad((lin(x) * wgt).sum(), lin.weight)
    want = cabi.sketch(kind, wgt, 96, base).T @ cabi.sketch(kind, x.detach(), 96, base, 1.0 / 96)
    assert torch.allclose(gw2, want, rtol=1e-4, atol=1e-3) and int(counter) == c0 + 3


def test_a_device_generator_still_determines_a_captured_sketch():
    """While the stream is capturing, a user-supplied DEVICE generator is not advanced (its offset belongs to torch's graph machinery), but it is
    not ignored either: the recorded seed kernel's base is derived from its initial seed, so the replayed product is the eager product with
    mix(base(initial_seed), counter), generators of different seeds give different matrices, the host's global RNG is untouched -- and the
    caller is told once (RuntimeWarning)."""
    import warnings
    import fewbit
    from fewbit_amd import linear
    x = torch.randn(512, 64, device=DEV, requires_grad=True)
    wgt = torch.randn(512, 32, device=DEV)
    results = {}
    linear._WARNED_CAPTURE_GENERATOR = False
    for seed in (11, 12):
        gen = torch.Generator(device=DEV).manual_seed(seed)
        lin = fewbit.RandomizedLinear(64, 32, proj_dim=96, matmul='rademacher', bias=False, device=DEV, generator=gen)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            torch.autograd.grad((lin(x) * wgt).sum(), lin.weight)
        torch.cuda.current_stream().wait_stream(side)
        counter = linear._replay_counter(torch.device(DEV))
        c0, offset, host_state = int(counter), gen.get_offset(), torch.random.get_rng_state()
        g = torch.cuda.CUDAGraph()
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            with torch.cuda.graph(g):
                gw, = torch.autograd.grad((lin(x) * wgt).sum(), lin.weight)
        if seed == 11:
            assert any(issubclass(w.category, RuntimeWarning) and 'device generator' in str(w.message) for w in caught)
        assert gen.get_offset() == offset and torch.equal(torch.random.get_rng_state(), host_state)
        g.replay()
        torch.cuda.synchronize()
        s = cabi.mix_sketch_seed(linear._mix64(seed, 0x6361707475726564), c0)
        want = cabi.sketch('rademacher', wgt, 96, s).T @ cabi.sketch('rademacher', x.detach(), 96, s, 1.0 / 96)
        assert torch.allclose(gw, want, rtol=1e-4, atol=1e-3), float((gw - want).abs().max())
        results[seed] = gw.clone()
    assert not torch.equal(results[11], results[12])
