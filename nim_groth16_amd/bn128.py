"""Host-side BN254 scalar-field helpers of the Python mirror (the reference's L1 facade,
groth16/bn128/{fields,io}.nim).  Only O(n) *scalar* bookkeeping lives here (byte <-> int conversion,
the Lagrange values of the fake setup); every group operation and every NTT/MSM runs on the GPU."""
from __future__ import annotations

primeP = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47   # fields.nim:36
primeR = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001   # fields.nim:37
_MONT = 1 << 256                                                               # io.nim:60-92
frMontR = _MONT % primeR                                                       # io.nim:91
frInvMontR = pow(_MONT, -1, primeR)                                            # io.nim:92
gen28 = 0x2a3c09f0a58a7e8500e0a7eb8ef62abc402d111e41112ed49bd61b6e725b19f0     # domain.nim:26


fpMontR = _MONT % primeP                                                       # io.nim:87
fpInvMontR = pow(_MONT, -1, primeP)                                            # io.nim:88


def fpFromMontBytes(b: bytes) -> int:
    return int.from_bytes(b, "little") * fpInvMontR % primeP


def fpToMontBytes(x: int) -> bytes:
    return (x % primeP * fpMontR % primeP).to_bytes(32, "little")


def frToMontBytes(x: int) -> bytes:
    """in-memory Fr (constantine limbs): Montgomery, little-endian"""
    return (x % primeR * frMontR % primeR).to_bytes(32, "little")


def frFromMontBytes(b: bytes) -> int:
    return int.from_bytes(b, "little") * frInvMontR % primeR


def frSeqToMontBytes(xs) -> bytes:
    return b"".join(frToMontBytes(x) for x in xs)


def frSeqToStdBytes(xs) -> bytes:
    """the .wtns layout (files/witness.nim:14): canonical little-endian, not Montgomery"""
    return b"".join((x % primeR).to_bytes(32, "little") for x in xs)


def frSeqFromMontBytes(buf: bytes):
    return [frFromMontBytes(buf[i:i + 32]) for i in range(0, len(buf), 32)]


def invFr(x: int) -> int:
    return pow(x, -1, primeR)


def batchInverseFr(xs):
    """fields.nim:163-174"""
    n = len(xs)
    pref = [1] * (n + 1)
    for i, x in enumerate(xs):
        pref[i + 1] = pref[i] * x % primeR
    inv = invFr(pref[n])
    out = [0] * n
    for i in range(n - 1, -1, -1):
        out[i] = inv * pref[i] % primeR
        inv = inv * xs[i] % primeR
    return out


def ceilingLog2(x: int) -> int:
    """misc.nim:43-47"""
    return -1 if x == 0 else (x - 1).bit_length()


def domainGen(log2n: int) -> int:
    """math/domain.nim:31-33"""
    return pow(gen28, 1 << (28 - log2n), primeR)
