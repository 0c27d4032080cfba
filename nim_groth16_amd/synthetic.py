"""Seeded synthetic circuits for the benchmark configurations:
a squaring chain  w_i * w_i = w_(i+1) - k_i  with m constraints.  Wires: [1, w_m (public output), w_0 .. w_(m-1)];
m = 2^k - 2 gives domainSize = 2^k (fake_setup.nim:203-206: ceilingLog2(m + npub + 1)).  Witness values are
full-width pseudo-random by construction (w_0 = 3, k_i from SplitMix64)."""
from __future__ import annotations

from . import bn128 as F
from .fake_setup import R1CS

R = F.primeR
_M64 = 0xFFFFFFFFFFFFFFFF


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & _M64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def fr(self) -> int:
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % R


def _fr_stream(seed: int, count: int):
    """== [SplitMix64(seed).fr() for _ in range(count)], vectorised (the state is seed + (i+1)*golden)."""
    import numpy as np
    idx = np.arange(1, 4 * count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    raw = z.astype("<u8").tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") % R for i in range(count)]


def squaringChain(m: int, seed: int = 4, w0: int = 3):
    """-> (R1CS, witness values as ints)."""
    ks = _fr_stream(seed, m)

    def wire(i):
        return i + 2 if i < m else 1
    cons = []
    w = [0] * (m + 1)
    w[0] = w0 % R
    for i in range(m):
        w[i + 1] = (w[i] * w[i] + ks[i]) % R
        cons.append(([(wire(i), 1)], [(wire(i), 1)], [(wire(i + 1), 1), (0, (-ks[i]) % R)]))
    witness = [1, w[m]] + w[:m]
    return R1CS(nWires=m + 2, nPubOut=1, nPubIn=0, nPrivIn=1, constraints=cons), witness
