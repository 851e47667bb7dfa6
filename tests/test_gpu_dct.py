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


@pytest.fixture(scope='module')
def ref():
    with np.load(GOLDEN / 'sampled_dct_ref.npz') as z:
        return {k: z[k].copy() for k in z.files}


def close(got, want, dtype):
    want = torch.as_tensor(want, dtype=torch.float64)
    err = (got.detach().cpu().double() - want).abs()
    tol = REL[dtype] * want.abs() + 3e-6 * float(want.abs().max())
    return bool((err <= tol).all()), float((err / tol.clamp_min(1e-300)).max())


@pytest.mark.parametrize('dtype', (torch.float32, torch.bfloat16, torch.float16))
def test_sampled_rows_equal_the_reference_run(ref, dtype):
    """9 shapes, 256 .. 65536 rows, ragged and odd feature counts, corner rows (0, rows/2, rows-1, the self-paired residue classes,
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
    for rows, features in ((1024, 130), (256, 64), (16384, 66), (32768, 34), (65536, 4)):
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
    assert cabi.sampled_dct_workspace_bytes(16384, 768, 3276) == 12 * 16384 * 256
    assert cabi.sampled_dct_workspace_bytes(16384, 70, 1) == 2 * 16384 * 256
    for rows in (48, 128, 3000, 131072):
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


@pytest.mark.parametrize('dtype', (torch.float32, torch.bfloat16))
def test_the_dct_layer_on_the_kernel_equals_the_layer_on_torch_fft(dtype):
    """linear_grp(matmul='dct') with 512 rows: the same generator state gives the same sampled rows on both paths, so the weight
    gradient of the native path equals that of the torch.fft formulation (which tests/test_gpu_linear.py pins to the reference's
    deterministic outputs); forward, input gradient and bias gradient are exact on both"""
    from fewbit_amd import linear
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 128, 40, generator=g).to(dtype).to(DEV)
    w = (torch.randn(24, 40, generator=g) * 0.3).to(dtype).to(DEV)
    b = torch.randn(24, generator=g).to(dtype).to(DEV)
    gy = torch.randn(4, 128, 24, generator=g).to(dtype).to(DEV)
    grads = {}
    for native in (True, False):
        prev = linear.use_native_sketch(native)
        try:
            xi, wi, bi = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
            gen = torch.Generator(device=DEV).manual_seed(99)
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
