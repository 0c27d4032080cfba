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


def mixedCircuit(m: int, seed: int = 4, zero_pct: int = 40, one_pct: int = 30, lin_pct: int = 0, w0: int = 3):
    """A circuit whose witness looks like a circom witness (SURVEY 8d config 2-ii / config 5's shape) instead of the
    squaring chain's full-width values.  Same wire layout as squaringChain -- [1, v_m (public), v_0 .. v_(m-1)],
    constraint i defines v_(i+1) -- with four kinds of constraint, drawn per row from SplitMix64(seed + 1000):
      zero_pct %  boolean wire holding 0:  b * b = b
      one_pct  %  boolean wire holding 1:  b * b = b
      lin_pct  %  linear step  (cur + k_i) * 1 = new  -- `cur` enters A only, so a wire that is only ever used in
                  linear steps is ABSENT FROM B and its pointsB1 / pointsB2 entries are the point at infinity (0,0),
                  as snarkjs keys have them for such wires (zkey.nim loads them as they are, curves.nim:95-98)
      the rest    squaring step  cur * cur = new - k_i  (full-width values)
    The last row is always a squaring step (it defines the public output).  -> (R1CS, witness values as ints)."""
    ks = _fr_stream(seed, m)
    kind = SplitMix64(seed + 1000)

    def wire(i):
        return i + 2 if i < m else 1
    cons = []
    v = [0] * (m + 1)
    v[0] = w0 % R
    cur = 0                                    # index of the latest full-width value
    for i in range(m):
        u = kind.next() % 100 if i + 1 < m else 100
        new = wire(i + 1)
        if u < zero_pct + one_pct:
            v[i + 1] = 0 if u < zero_pct else 1
            cons.append(([(new, 1)], [(new, 1)], [(new, 1)]))
        elif u < zero_pct + one_pct + lin_pct:
            v[i + 1] = (v[cur] + ks[i]) % R
            cons.append(([(wire(cur), 1), (0, ks[i])], [(0, 1)], [(new, 1)]))
            cur = i + 1
        else:
            v[i + 1] = (v[cur] * v[cur] + ks[i]) % R
            cons.append(([(wire(cur), 1)], [(wire(cur), 1)], [(new, 1), (0, (-ks[i]) % R)]))
            cur = i + 1
    witness = [1, v[m]] + v[:m]
    return R1CS(nWires=m + 2, nPubOut=1, nPubIn=0, nPrivIn=1, constraints=cons), witness
