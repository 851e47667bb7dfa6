"""The generator behind the random-projection kernel, without a GPU: Philox4x32-10 and xoshiro128++ of the library (host entry
points of the C-ABI) and of the numpy model against published known-answer vectors, and the statistics of the model's matrices."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import sketch_reference as ref
from fewbit_amd import cabi
from helpers import ROOT

# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key, expected)
KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff, ) * 4, (0xffffffff, ) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answers_library_and_model():
    for ctr, key, want in KAT:
        assert cabi.philox4x32(ctr, key) == want
        got = ref.philox4x32(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want
    # the model on arrays == the library call by call
    rng = np.random.default_rng(0)
    c = rng.integers(0, 2**32, size=(4, 50), dtype=np.uint64)
    got = ref.philox4x32(c[0], c[1], c[2], c[3], 0x12345678, 0x9abcdef0)
    for n in range(50):
        assert cabi.philox4x32(tuple(int(c[k][n]) for k in range(4)), (0x12345678, 0x9abcdef0)) == tuple(int(g[n]) for g in got)


def test_model_round_loop_against_the_seven_round_vectors_too():
    """Random123 kat_vectors, philox4x32 with 7 rounds: a second published fixed point for the model's round function and key
    schedule (the kernels use 10 rounds; EXPERIMENTS.md 3.1 records the 7-round experiment)"""
    for ctr, key, want in (((0, 0, 0, 0), (0, 0), (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
                           ((0xffffffff, ) * 4, (0xffffffff, ) * 2, (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
                           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a))):
        got = ref.philox4x32(*[np.array([c]) for c in ctr], *key, rounds=7)
        assert tuple(int(g[0]) for g in got) == want


def test_xoshiro128pp_known_answers_library_and_model():
    """xoshiro128++ 1.0: the outputs of Blackman & Vigna's reference xoshiro128plusplus.c from the state {1, 2, 3, 4} (the vector
    the rand_xoshiro crate's tests carry); the first two are checkable by hand: rotl(1 + 4, 7) + 1 = 641, then the state is
    {7, 0, 1026, 12288} and rotl(7 + 12288, 7) + 7 = 1573767"""
    want = [641, 1573767, 3222811527, 3517856514, 836907274, 4247214768, 3867114732, 1355841295, 495546011, 621204420]
    out, state = cabi.xoshiro128pp((1, 2, 3, 4), 10)
    assert list(out) == want
    more, _ = cabi.xoshiro128pp(state, 5)                                  # the state handed back continues the stream
    assert list(cabi.xoshiro128pp((1, 2, 3, 4), 15)[0]) == want + list(more)
    st = [np.array([v, v + 10], dtype=np.uint64) for v in (1, 2, 3, 4)]
    got = []
    for _ in range(10):
        w, st = ref.xoshiro128pp(st)
        got.append(int(w[0]))
    assert got == want
    assert int(w[1]) == cabi.xoshiro128pp((11, 12, 13, 14), 10)[0][-1]     # the model on arrays == the library lane by lane


def test_gaussian_words_are_the_streams_the_kernel_header_defines():
    """S[i][r] for the Gaussian sketch: word 2 s + q % 2 of stream q // 2 of (row i, block r // 256, octet parity (r // 8) % 2),
    the stream seeded by philox(i, 2 (r // 256) + h, q // 2, 2) -- spelled out with the library's host generators"""
    seed = 0x0123456789abcdef
    key = (seed & 0xffffffff, seed >> 32)
    for i, r in ((0, 0), (5, 7), (5, 8), (77, 255), (77, 256), (3, 1000), (2**31 + 1, 2**33 + 8 * 37 + 5)):
        s, h, q = (r % 256) // 16, (r // 8) % 2, (r % 8) // 2
        state = cabi.philox4x32((i & 0xffffffff, (2 * (r // 256) + h) & 0xffffffff, q // 2, 2), key)
        n = 2 * s + q % 2
        want = cabi.xoshiro128pp(state, n + 1)[0][n]
        assert int(ref.gaussian_words(seed, np.array([[i]]), np.array([[r]]))[0, 0]) == want, (i, r)


def test_seed_derivation_of_captured_launches_on_the_host():
    """fewbit_hip_sketch_mix_seed (what the recorded seed kernel computes from its counter): splitmix64's finaliser"""
    M = 2**64 - 1

    def splitmix(base, count):
        x = (base + (count + 1) * 0x9E3779B97F4A7C15) & M
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
        return x ^ (x >> 31)

    rng = np.random.default_rng(1)
    for base, count in [(0, 0), (M, M), (1, 0), (0, 1)] + [(int(a), int(b)) for a, b in rng.integers(0, 2**63, size=(50, 2))]:
        assert cabi.mix_sketch_seed(base, count) == splitmix(base, count)
    # splitmix64's own first output for state 0 (Vigna's reference implementation): 0xe220a8397b1dcdaf
    assert cabi.mix_sketch_seed(0, 0) == 0xe220a8397b1dcdaf
    assert len({cabi.mix_sketch_seed(5, c) for c in range(1000)}) == 1000


def test_model_matrices_have_the_right_moments_and_are_functions_of_the_seed():
    S = ref.rademacher(7, 96, 2048)
    assert set(S.unique().tolist()) == {-1.0, 1.0}
    assert abs(float(S.mean())) < 0.01 and abs(float((S[:48] * S[48:]).mean())) < 0.01          # rows uncorrelated
    assert abs(float((S[:, :-1] * S[:, 1:]).mean())) < 0.01                                       # neighbours uncorrelated
    G = ref.gaussian(7, 96, 2048, rounded=False)
    assert abs(float(G.mean())) < 0.01 and abs(float(G.var()) - 1.0) < 0.02
    kurt = float((G**4).mean() / G.var()**2)
    assert abs(kurt - 3.0) < 0.1
    assert abs(float((G[:, 0::2] * G[:, 1::2]).mean())) < 0.01                                   # the two halves of a Box-Muller pair
    # consecutive words of a stream (elements 0,1 against 2,3 of an octet), the two streams (0..3 against 4..7), consecutive
    # steps of a block (columns r and r + 16) and the two octet parities (r and r + 8): uncorrelated, also in their squares
    for a, b in ((G[:, 0::8], G[:, 2::8]), (G[:, 1::8], G[:, 5::8]), (G[:, :-16], G[:, 16:]), (G[:, :-8], G[:, 8:]), (G[:-1], G[1:])):
        assert abs(float((a * b).mean())) < 0.012 and abs(float((a * a * b * b).mean()) - 1.0) < 0.04
    # windows of the same matrix agree with the whole; another seed gives another matrix
    assert torch.equal(ref.rademacher(7, 10, 300, row0=5, col0=250), ref.rademacher(7, 96, 2048)[5:15, 250:550])
    assert torch.equal(ref.gaussian(7, 10, 300, row0=5, col0=250), ref.gaussian(7, 96, 2048)[5:15, 250:550])
    assert not torch.equal(ref.rademacher(8, 96, 2048), S)
    # E[S^T S] = proj * I: the property the estimator's unbiasedness rests on
    S = ref.rademacher(3, 4096, 24)
    assert float((S.T @ S / 4096 - torch.eye(24)).abs().max()) < 0.07


def _plan(dist, dtype, rows, features, proj):
    """fewbit_hip_sketch_describe through ctypes (the host-side planner: no GPU needed, an absent device counts as 256 CUs)"""
    L = cabi.lib()
    buf = ctypes.create_string_buffer(1024)
    assert L.fewbit_hip_sketch_describe(cabi.SKETCH_DISTS.index(dist), cabi.DTYPES[dtype], rows, features, proj, buf, 1024) == 0
    return json.loads(buf.value.decode())


def test_data_path_policy_of_the_sketch_without_a_gpu():
    """which of the data paths a call takes: Gaussian S from memory for 16-bit input wider than one tile and for fp32
    input of 2048 features or more, the fused kernel for narrower fp32 input (FEWBIT_SKETCH_MATERIALISE=1 overrides) and for
    Rademacher; bf16 partial sums for sliced bf16 operands; the workspace is the sum of its parts"""
    g = _plan('gaussian', torch.bfloat16, 16384, 768, 3276)
    assert 'from memory' in g['kernel'] and g['grid'] == [3, 13, 6] and g['partial_sums'] == 'bf16'
    frag = (13 * 8 * 64 * 16 + 4) * 1024
    assert g['s_fragment_bytes'] == frag and g['workspace_bytes'] == 6 * 3276 * 768 * 2 + frag          # (the partial sums end on a 256-byte boundary here)
    assert 'from memory' in _plan('gaussian', torch.float16, 16384, 3072, 3276)['kernel']
    assert _plan('gaussian', torch.float16, 16384, 3072, 3276)['partial_sums'] == 'fp32'                 # fp16: range
    assert 'from memory' not in _plan('gaussian', torch.bfloat16, 16384, 256, 3276)['kernel']            # one column tile: nothing to share
    assert 'from memory' not in _plan('rademacher', torch.bfloat16, 16384, 3072, 3276)['kernel']
    assert _plan('gaussian', torch.bfloat16, 2**21, 768, 2**10)['s_fragment_bytes'] == 0                 # 4 GiB of fragments: the fused kernel
    f = _plan('gaussian', torch.float32, 16384, 768, 3276)                                                # fp32 input: rounded to bf16 first, fused kernel
    assert f['converted_to_bf16_first'] is True and 'from memory' not in f['kernel'] and f['partial_sums'] == 'bf16'
    assert f['workspace_bytes'] == -(-(f['grid'][2] * 3276 * 768 * 2) // 256) * 256 + 16384 * 768 * 2
    w = _plan('gaussian', torch.float32, 16384, 3072, 3276)                                               # ... a wide layer: S from memory
    assert w['converted_to_bf16_first'] is True and 'from memory' in w['kernel'] and w['s_fragment_bytes'] == frag
    assert w['workspace_bytes'] == -(-(-(-(w['grid'][2] * 3276 * 3072 * 2) // 256) * 256 + 16384 * 3072 * 2) // 256) * 256 + frag
    assert 'from memory' not in _plan('gaussian', torch.float32, 16384, 2040, 3276)['kernel'] and 'from memory' in _plan('gaussian', torch.float32, 16384, 2048, 3276)['kernel']
    small = _plan('rademacher', torch.float32, 16384, 768, 200)                                           # fp32 operand staged in the kernel
    assert small['converted_to_bf16_first'] is False and small['partial_sums'] == 'fp32'
    code = ("import ctypes, json, torch; from fewbit_amd import cabi; L = cabi.lib(); b = ctypes.create_string_buffer(1024); "
            "L.fewbit_hip_sketch_describe(1, 0, 16384, 768, 3276, b, 1024); print(json.loads(b.value.decode())['kernel'])")
    for env, expect in (('1', True), ('0', False)):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, FEWBIT_SKETCH_MATERIALISE=env, PYTHONPATH=str(ROOT)), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and ('from memory' in r.stdout) == expect, (env, r.stdout, r.stderr[-500:])


def test_rows_of_a_seed_are_the_documented_function_and_uniform():
    """idx[j] = 16-bit half j % 8 of Philox4x32-10((j / 8, 0, 0, 3), seed), mod rows (include/fewbit_hip.h): the host evaluation against the independent
    Philox of tests/sketch_reference.py (itself pinned by the Random123 known answers above), and a chi-square of 2^20 samples over 256 rows (255 degrees of freedom: 99.99 % point 347)"""
    seed = 0x0123456789abcdef
    idx = cabi.sampled_rows(seed, 4096, 1003)
    key = (seed & 0xffffffff, seed >> 32)
    want = [(int(w) >> s) & 4095 for q in range(126) for w in ref.philox4x32(q, 0, 0, 3, *key) for s in (0, 16)][:1003]
    assert idx.tolist() == want
    assert cabi.sampled_rows(seed, 4096, 0).numel() == 0
    counts = torch.bincount(cabi.sampled_rows(5, 256, 1 << 20), minlength=256).double()
    assert float(((counts - 4096.0) ** 2 / 4096.0).sum()) < 347.0
    # rows = 3 x 2^k: 32-bit words, idx[j] = (word j % 4 of call j / 4) * rows >> 32
    idx = cabi.sampled_rows(seed, 12288, 1003)
    want = [(int(w) * 12288) >> 32 for q in range(251) for w in ref.philox4x32(q, 0, 0, 3, *key)][:1003]
    assert idx.tolist() == want and 0 <= min(want) and max(want) < 12288
    counts = torch.bincount(cabi.sampled_rows(6, 768, 3 << 20), minlength=768).double()
    assert float(((counts - 4096.0) ** 2 / 4096.0).sum()) < 920.0            # 767 degrees of freedom: 99.99 % point 917
    # rows = 2^17, 2^18: 32-bit words too
    idx = cabi.sampled_rows(seed, 262144, 1003)
    want = [(int(w) * 262144) >> 32 for q in range(251) for w in ref.philox4x32(q, 0, 0, 3, *key)][:1003]
    assert idx.tolist() == want
    assert int(cabi.sampled_rows(seed, 20480, 4000).max()) < 20480
    for rows in (1000, 384, 1792, 81920, 98304, 524288, 128):
        with pytest.raises(cabi.FewbitHipError):
            cabi.sampled_rows(1, rows, 4)
