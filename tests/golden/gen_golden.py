#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by RUNNING THE REFERENCE in the build container.

Run from the repo root, only where /root/reference exists:

    make -C oracle ref && python tests/golden/gen_golden.py            (everything)
    python tests/golden/gen_golden.py fullsize                         (only the full-size digests)

What is executed is the reference itself, where it lies:
  * oracle/_ref/libfewbit_ref.so  = fewbit/fewbit.cc + fewbit/cpu/gelu.cc + fewbit/cpu/codec.cc
    (``torch.ops.fewbit.quantize`` / ``quantize_backward``, fewbit/cpu/gelu.cc:7-45)
  * oracle/_ref/libcodec_ref.so   = fewbit/cpu/codec.h (``Deflate``/``Inflate``) for the 1-bit
    width the native op cannot reach (SURVEY 2.2 defect 5)
  * the reference's python package, imported through a symlink farm in a temp dir, for the
    API-surface fixture (signatures / reprs / table casts).
Only inputs and outputs (data) are written; no reference text is stored.
This script does not import fewbit_amd or oracle.
"""
import ctypes
import inspect
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
REF = Path('/root/reference')
OUT = Path(__file__).resolve().parent
DTYPES = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}


def raw(t: torch.Tensor) -> np.ndarray:
    """bit-exact numpy view (16-bit floats as uint16, fp32 as float32)."""
    t = t.contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    return t.numpy().copy()


def specials(borders: torch.Tensor, dtype) -> torch.Tensor:
    """NaN (both signs), +-inf, +-0, every border and its neighbours in `dtype`, +-100, +-200."""
    b = borders.to(dtype)
    vals = [torch.tensor([float('nan'), float('inf'), -float('inf'), 0.0, -0.0, 100.0, -100.0, 200.0, -200.0,
                          1e-30, -1e-30, 5.0, -5.0, 9.0, -9.0]).to(dtype)]
    if dtype == torch.float32:
        nb = b.view(torch.int32)
        vals += [b, (nb + 1).view(dtype), (nb - 1).view(dtype)]
        neg_nan = torch.tensor([-1], dtype=torch.int32).view(dtype)        # 0xffffffff
    else:
        nb = b.view(torch.int16)
        vals += [b, (nb + 1).view(dtype), (nb - 1).view(dtype)]
        neg_nan = torch.tensor([-1], dtype=torch.int16).view(dtype)        # 0xffff
    vals.append(neg_nan)
    return torch.cat(vals)


def make_inputs(n: int, dtype, borders, seed: int):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, generator=g) * 1.5).to(dtype)
    gy = torch.randn(n, generator=g).to(dtype)
    if n >= 64:
        sp = specials(borders, dtype)
        m = min(sp.numel(), n)
        x[:m] = sp[:m]
    return x, gy


def gen_quantize(tables):
    torch.ops.load_library(str(ROOT / 'oracle/_ref/libfewbit_ref.so'))
    out = {}
    sizes = (1, 7, 8, 9, 64, 65, 257, 1001)
    cases = [('gelu', k, dt, sizes) for k in (2, 3, 4) for dt in DTYPES]
    cases += [('silu', 2, 'f16', (1001,)), ('silu', 4, 'f16', (1001,)), ('tanh', 3, 'f32', (257,))]
    for name, k, dt, ns in cases:
        dtype = DTYPES[dt]
        borders = torch.tensor(tables[f'{name}{k:02d}-borders'])
        levels = torch.tensor(tables[f'{name}{k:02d}-levels'])
        # what fewbit/functional/activations.py:210-213 hands to the op
        b_in = borders.to(dtype)[1:-1].contiguous()
        l_in = levels.to(dtype)
        for n in ns:
            x, gy = make_inputs(n, dtype, borders[1:-1], seed=1000 * k + n)
            y, state = torch.ops.fewbit.quantize(x, b_in)   # y is ATen gelu whatever the table
            gx = torch.ops.fewbit.quantize_backward(gy, state, l_in)
            key = f'{name}{k:02d}_{dt}_{n}'
            out[key + '_x'] = raw(x)
            out[key + '_gy'] = raw(gy)
            out[key + '_state'] = state.numpy().copy()
            out[key + '_gx'] = raw(gx)
            if name == 'gelu':
                out[key + '_y'] = raw(y)
        out[f'{name}{k:02d}_{dt}_borders'] = raw(b_in)
        out[f'{name}{k:02d}_{dt}_levels'] = raw(l_in)
    np.savez_compressed(OUT / 'quantize_ref.npz', **out)
    print('quantize_ref.npz', len(out), 'arrays')


def gen_codec():
    L = ctypes.CDLL(str(ROOT / 'oracle/_ref/libcodec_ref.so'))
    L.ref_deflate_u8.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int32]
    L.ref_inflate_u8.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int32]
    rng = np.random.default_rng(42)
    out = {}
    for k in range(1, 9):
        for n in (1, 5, 8, 11, 64, 256, 1001):
            codes = rng.integers(0, 1 << k, n).astype(np.int32)
            nbytes = (k * n + 7) // 8
            buf = np.zeros(nbytes + 1, np.uint8)
            L.ref_deflate_u8(codes.ctypes.data, n, buf.ctypes.data, k)
            back = np.zeros(n, np.int32)
            L.ref_inflate_u8(back.ctypes.data, n, buf.ctypes.data, k)
            assert (back == codes).all()
            out[f'k{k}_n{n}_codes'] = codes.astype(np.uint8)
            out[f'k{k}_n{n}_bytes'] = buf[:nbytes].copy()
    # 1-bit relu end to end: bit rule of fewbit/cuda/codec.cu:412-425 + reference Deflate(…, 1)
    g = torch.Generator().manual_seed(7)
    for dt, dtype in DTYPES.items():
        n = 1001
        x = torch.randn(n, generator=g).to(dtype)
        x[:6] = torch.tensor([0.0, -0.0, float('inf'), -float('inf'), float('nan'), 1e-30]).to(dtype)
        gy = torch.randn(n, generator=g).to(dtype)
        bits = (x.float() > 0).to(torch.int32).numpy()          # `x <= 0 -> 0 else 1`: NaN -> 1
        bits[4] = 1
        buf = np.zeros((n + 7) // 8 + 1, np.uint8)
        L.ref_deflate_u8(bits.ctypes.data, n, buf.ctypes.data, 1)
        out[f'relu01_{dt}_x'] = raw(x)
        out[f'relu01_{dt}_gy'] = raw(gy)
        out[f'relu01_{dt}_state'] = buf[:(n + 7) // 8].copy()
        out[f'relu01_{dt}_y'] = raw(torch.relu(x))
    np.savez_compressed(OUT / 'codec_ref.npz', **out)
    print('codec_ref.npz', len(out), 'arrays')


def gen_api(tables):
    """Import the reference python package (symlink farm + the CPU op library) and record its surface."""
    tmp = Path(tempfile.mkdtemp(prefix='fbref_'))
    pkg = tmp / 'fewbit'
    for src in (REF / 'fewbit').rglob('*'):
        rel = src.relative_to(REF / 'fewbit')
        if src.is_dir():
            (pkg / rel).mkdir(parents=True, exist_ok=True)
        else:
            (pkg / rel).parent.mkdir(parents=True, exist_ok=True)
            os.symlink(src, pkg / rel)
    os.symlink(ROOT / 'oracle/_ref/libfewbit_ref.so', pkg / 'libfewbit.so')
    sys.path.insert(0, str(tmp))
    import fewbit  # the reference
    assert Path(fewbit.__file__).parent == pkg
    api = {'functional': {}, 'modules': {}, 'tables_cast': {}}
    from fewbit.functional import activations as fa
    for name in fa.STEPWISE + fa.CONTINOUS:
        fn = getattr(fewbit.functional, name, None) or getattr(fa, name)
        try:
            api['functional'][name] = str(inspect.signature(fn))
        except (TypeError, ValueError):
            api['functional'][name] = None
    from fewbit.modules import activations as ma
    for name in ma.__all__:
        cls = getattr(fewbit, name, None) or getattr(ma, name)
        entry = {'init': str(inspect.signature(cls.__init__))}
        try:
            if name == 'Threshold':
                entry['repr'] = repr(cls(1.0, 3.0, bits=3))
            elif name == 'Stepwise':
                entry['repr'] = None
            else:
                entry['repr'] = repr(cls(bits=3))
                entry['repr_default'] = repr(cls())
        except Exception as e:  # noqa: BLE001
            entry['repr'] = f'ERR {type(e).__name__}'
        api['modules'][name] = entry
    for name, bits in (('gelu', 3), ('silu', 2), ('silu', 4), ('gelu', 1)):
        for dt, dtype in DTYPES.items():
            b, l = fa.store.get(name, bits, 'cpu', dtype)
            api['tables_cast'][f'{name}{bits:02d}_{dt}'] = {
                'borders': raw(b).view(np.uint32 if dt == 'f32' else np.uint16).tolist(),
                'levels': raw(l).view(np.uint32 if dt == 'f32' else np.uint16).tolist(),
            }
    api['store_len'] = len(fa.store)
    api['store_keys'] = sorted(f'{k[0]}{k[1]:02d}' for k, _ in fa.store.items())
    # error behaviour of the dispatcher (fewbit/functional/activations.py:196-202, :60-61)
    errs = {}
    x = torch.zeros(4)
    try:
        fewbit.functional.gelu(x, bits=3, borders=torch.zeros(3), values=torch.zeros(4))
    except Exception as e:  # noqa: BLE001
        errs['bits_and_custom'] = type(e).__name__
    try:
        fewbit.functional.gelu(x, bits=7)
    except Exception as e:  # noqa: BLE001
        errs['unknown_bits'] = type(e).__name__
    api['errors'] = errs
    (OUT / 'api_surface.json').write_text(json.dumps(api, indent=1, sort_keys=True) + '\n')
    print('api_surface.json', {k: len(v) if hasattr(v, '__len__') else v for k, v in api.items()})


def gen_fullsize(tables):
    """The reference itself on the full BASELINE-size tensors (C2, C3 k=2/k=4, the C4 per-GPU shard): only SHA-256
    digests of x, gy, the packed state and gx are committed (tests/golden/fullsize_digests.json); the GPU test
    regenerates x/gy with the same seeded recipe (tests/helpers.py:full_size_inputs), checks their digests, and compares
    the digests of the HIP kernels' state and gx."""
    sys.path.insert(0, str(ROOT / 'tests'))
    from helpers import FULL_SIZE_CASES, full_size_inputs, sha256_of
    torch.ops.load_library(str(ROOT / 'oracle/_ref/libfewbit_ref.so'))
    doc = {'recipe': 'tests/helpers.py:full_size_inputs; state/gx from torch.ops.fewbit.quantize / quantize_backward of '
                     'oracle/_ref/libfewbit_ref.so (the reference compiled as is); sha256 over the raw little-endian bytes',
           'torch': torch.__version__, 'cases': {}}
    for case, (name, bits, dt, rows, cols) in FULL_SIZE_CASES.items():
        x, gy, b, l = full_size_inputs(case, tables)
        _, state = torch.ops.fewbit.quantize(x, b)
        gx = torch.ops.fewbit.quantize_backward(gy, state, l)
        assert state.numel() == bits * (rows * cols // 8)
        doc['cases'][case] = {'function': name, 'bits': bits, 'dtype': dt, 'shape': [rows, cols], 'x': sha256_of(x),
                              'gy': sha256_of(gy), 'state': sha256_of(state), 'gx': sha256_of(gx),
                              'state_bytes': state.numel(), 'state_byte_sum': int(state.sum(dtype=torch.int64))}
        print(case, doc['cases'][case]['state'][:16], doc['cases'][case]['gx'][:16])
    (OUT / 'fullsize_digests.json').write_text(json.dumps(doc, indent=1, sort_keys=True) + '\n')


def gen_c4(tables):
    """BASELINE config 4 in full: the reference on each of the 4 WHOLE 16384x4096 bf16 tensors; committed are the SHA-256
    digests of the 8 shards' slices (x, gy, packed state, gx) -- tests/golden/c4_digests.json.  tests/test_gpu_multi.py
    regenerates the inputs (tests/helpers.py:c4_tensor_inputs), runs shard r on device r and compares digests."""
    sys.path.insert(0, str(ROOT / 'tests'))
    sys.path.insert(0, str(ROOT))
    from helpers import C4_BITS, C4_COLS, C4_ROWS, C4_TENSORS, c4_shard, c4_tensor_inputs, sha256_of
    torch.ops.load_library(str(ROOT / 'oracle/_ref/libfewbit_ref.so'))
    doc = {'recipe': 'tests/helpers.py:c4_tensor_inputs / c4_shard; state and gx from torch.ops.fewbit.quantize / quantize_backward of '
                     'oracle/_ref/libfewbit_ref.so (the reference compiled as is) run on each WHOLE tensor, then sliced at the '
                     'shard offsets of fewbit_amd.sharding; sha256 over the raw little-endian bytes',
           'config': f'{C4_TENSORS} x ({C4_ROWS}x{C4_COLS}) bf16, gelu {C4_BITS} bits, 8 shards: tensor t half h -> GPU 2t+h',
           'torch': torch.__version__, 'tensors': {}, 'shards': {}}
    for t in range(C4_TENSORS):
        x, gy, b, l = c4_tensor_inputs(t, tables)
        _, state = torch.ops.fewbit.quantize(x, b)
        gx = torch.ops.fewbit.quantize_backward(gy, state, l)
        doc['tensors'][str(t)] = {'x': sha256_of(x), 'gy': sha256_of(gy), 'state': sha256_of(state), 'gx': sha256_of(gx)}
        for h in range(2):
            r = 2 * t + h
            tt, (e0, e1), (s0, s1) = c4_shard(r)
            assert tt == t
            doc['shards'][str(r)] = {'tensor': t, 'elements': [e0, e1], 'state_bytes': [s0, s1], 'x': sha256_of(x[e0:e1]),
                                     'gy': sha256_of(gy[e0:e1]), 'state': sha256_of(state[s0:s1]), 'gx': sha256_of(gx[e0:e1]),
                                     'state_byte_sum': int(state[s0:s1].sum(dtype=torch.int64))}
            print('shard', r, doc['shards'][str(r)]['state'][:16], doc['shards'][str(r)]['gx'][:16])
    (OUT / 'c4_digests.json').write_text(json.dumps(doc, indent=1, sort_keys=True) + '\n')


def main():
    assert REF.exists(), 'the reference tree is only present in the build container'
    with np.load(REF / 'fewbit/data/builtin.npz') as z:
        tables = {k: z[k].copy() for k in z.files}
    if len(sys.argv) > 1 and sys.argv[1] == 'fullsize':
        gen_fullsize(tables)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'c4':
        gen_c4(tables)
        return
    gen_quantize(tables)
    gen_codec()
    gen_api(tables)
    gen_fullsize(tables)
    gen_c4(tables)


if __name__ == '__main__':
    main()
