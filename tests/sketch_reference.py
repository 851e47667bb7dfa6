"""Host model of the random matrix S behind fewbit_hip_sketch (fewbit_amd/csrc/fewbit_sketch.hip): the same formulas,
evaluated with numpy -- a pure function of (seed, row, column).  Test infrastructure (the checker), never the product path.

    philox4x32(c0..c3, k0, k1)      Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
                                    SC'11), vectorised over arrays of counters
    rademacher(seed, rows, cols)    +-1 matrix S[rows x cols]
    xoshiro128pp(state)             xoshiro128++ 1.0 (Blackman & Vigna 2019, reference xoshiro128plusplus.c), vectorised
    gaussian(seed, rows, cols, dt)  N(0,1) by Box-Muller on 16-bit uniforms, rounded to the operand dtype `dt`; the 32-bit words
                                    come from two xoshiro128++ streams per (row, 256-column block, octet parity), each seeded
                                    by one Philox call
"""
import numpy as np
import torch

M32 = np.uint64(0xffffffff)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & M32 for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = np.uint64(k0 & 0xffffffff), np.uint64(k1 & 0xffffffff)
    for _ in range(rounds):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        c1, c3, c0, c2 = p1 & M32, p0 & M32, n0, n2
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return c0, c1, c2, c3


def _key(seed):
    seed &= 0xffffffffffffffff
    return seed & 0xffffffff, seed >> 32


def rademacher(seed: int, nrows: int, ncols: int, row0: int = 0, col0: int = 0) -> torch.Tensor:
    i = (row0 + np.arange(nrows, dtype=np.uint64))[:, None]
    r = (col0 + np.arange(ncols, dtype=np.uint64))[None, :]
    s, h, j = (r % 256) // 16, (r // 8) % 2, r % 8
    words = philox4x32(i, 2 * (r // 256) + h, 0, 0, *_key(seed))
    word = np.choose((s // 4).astype(np.int64) + np.zeros(i.shape, dtype=np.int64), words)
    bit = (word >> (np.where(j % 2 == 1, 31, 15).astype(np.uint64) - (np.uint64(4) * (s % 4) + j // 2))) & np.uint64(1)
    return torch.from_numpy(np.where(bit == 1, -1.0, 1.0).astype(np.float32))


def xoshiro128pp(state):
    """one step for arrays of states (a list of four uint64 arrays holding 32-bit values): returns (output, new state)"""
    s0, s1, s2, s3 = state

    def rotl(x, k):
        return ((x << np.uint64(k)) | (x >> np.uint64(32 - k))) & M32

    result = (rotl((s0 + s3) & M32, 7) + s0) & M32
    t = (s1 << np.uint64(9)) & M32
    s2 = s2 ^ s0
    s3 = s3 ^ s1
    s1 = s1 ^ s2
    s0 = s0 ^ s3
    s2 = s2 ^ t
    s3 = rotl(s3, 11)
    return result, [s0, s1, s2, s3]


def gaussian_words(seed: int, i, r):
    """the 32-bit word behind S[i][r] (arrays, broadcast against each other): with b = r // 256, s = (r % 256) // 16,
    h = (r // 8) % 2, q = (r % 8) // 2: output number 2 s + q % 2 of the xoshiro128++ stream whose state is
    philox(i, 2 b + h, q // 2, 2) -- stream q // 2 feeds operand dwords 2 (q // 2) and 2 (q // 2) + 1 of every MFMA step of the block"""
    i, r = np.broadcast_arrays(np.asarray(i, dtype=np.uint64), np.asarray(r, dtype=np.uint64))
    s, h, q = (r % 256) // 16, (r // 8) % 2, (r % 8) // 2
    state = list(philox4x32(i, 2 * (r // 256) + h, q // 2, 2, *_key(seed)))
    n = (2 * s + q % 2).astype(np.int64)
    word = np.zeros(i.shape, dtype=np.uint64)
    for k in range(int(n.max()) + 1 if n.size else 0):
        out, state = xoshiro128pp(state)
        word = np.where(n == k, out, word)
    return word


def gaussian(seed: int, nrows: int, ncols: int, dtype: torch.dtype = torch.bfloat16, row0: int = 0, col0: int = 0,
             rounded: bool = True) -> torch.Tensor:
    i = (row0 + np.arange(nrows, dtype=np.uint64))[:, None]
    r = (col0 + np.arange(ncols, dtype=np.uint64))[None, :]
    j = r % 8
    w = gaussian_words(seed, i, r)
    u1 = ((w & np.uint64(0xffff)).astype(np.float64) + 0.5) / 65536.0
    u2 = (w >> np.uint64(16)).astype(np.float64) / 65536.0
    rad = np.sqrt(-2.0 * np.log(u1))
    z = np.where(j % 2 == 0, rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2))
    t = torch.from_numpy(z)
    if not rounded:
        return t
    op = torch.float16 if dtype == torch.float16 else torch.bfloat16        # fp32 inputs run on the bf16 pipe
    return t.to(torch.float32).to(op).to(torch.float32)


def matrix(dist: str, seed: int, nrows: int, ncols: int, dtype: torch.dtype = torch.bfloat16, row0: int = 0, col0: int = 0):
    if dist == 'rademacher':
        return rademacher(seed, nrows, ncols, row0, col0)
    return gaussian(seed, nrows, ncols, dtype, row0, col0)
