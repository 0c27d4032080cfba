"""groth16/files/zkey.nim: the snarkjs proving key.  Point sections (3, 5-9) and the spec points hold
little-endian Montgomery (R = 2^256) coordinates -- byte for byte the in-memory G1/G2 layout the GPU library
takes (bn128/io.nim:103-131), so they are passed through unparsed.  Section 4 coefficients are *doubly*
Montgomery encoded (zkey.nim:57, io.nim:134-139)."""
from __future__ import annotations

import struct

import numpy as np

from .. import bn128 as F
from ..zkey_types import GrothHeader, ProverPoints, Snarkjs, SpecPoints, ZKey
from .container import parseContainer, parsePrimeField, writeContainer

_R = F.primeR


def checkZKeyPoints(zk: ZKey, ctx=None) -> None:
    """The reference asserts that every loaded point lies on its curve (mkG1/mkG2, curves.nim:95-107).  Here the
    check is one GPU pass per section, run on request (parseZKey(..., check=True))."""
    from .._lib import default_context
    ctx = ctx or default_context()
    sp, pp = zk.specPoints, zk.pPoints
    for name, group, buf in (("spec G1", 1, sp.alpha1 + sp.beta1 + sp.delta1), ("spec G2", 2, sp.beta2 + sp.gamma2 + sp.delta2),
                             ("pointsIC", 1, zk.pointsIC), ("pointsA1", 1, pp.pointsA1), ("pointsB1", 1, pp.pointsB1),
                             ("pointsB2", 2, pp.pointsB2), ("pointsC1", 1, pp.pointsC1), ("pointsH1", 1, pp.pointsH1)):
        bad = ctx.points_check(group, buf)
        assert bad is None, f"mkG{group}: {name}[{bad}] is not a G{group} curve point"


def parseZKey(fname: str, check: bool = False, ctx=None, rawCoeffs: bool = False) -> ZKey:
    """zkey.nim:241-246.  rawCoeffs: keep the coefficient section as it lies in the file (ZKey.coeffsSection4, handed to
    the GPU unparsed by loadProvingKey -> g16_pkey_create_zkey) instead of un-Montgomerying every entry on the host
    (io.nim:134-139): what a large key wants -- a Poseidon-Merkle circuit of 2^20 constraints has ~10^7 entries."""
    sec = parseContainer("zkey", 1, fname)
    one = lambda i: sec[i][0]                                            # noqa: E731
    s1 = one(1)                                                          # zkey.nim:104-107
    assert len(s1) == 4, "unexpected section length"
    assert struct.unpack_from("<I", s1, 0)[0] == 1, "expecting `.zkey` file for a Groth16 prover"
    s2 = one(2)                                                          # zkey.nim:114-165
    n8p, p, pos = parsePrimeField(s2, 0)
    n8r, r, pos = parsePrimeField(s2, pos)
    assert len(s2) == 2 * 4 + n8p + n8r + 3 * 4 + 3 * 64 + 3 * 128, "unexpected section length"
    assert n8p == 32 and n8r == 32, "expecting 256 bit primes"
    assert p == F.primeP and r == F.primeR, "expecting the alt-bn128 curve"
    nvars, npubs, domsiz = struct.unpack_from("<III", s2, pos)
    pos += 12
    log2siz = F.ceilingLog2(domsiz)
    assert (1 << log2siz) == domsiz, "domain size should be a power of two"
    zk = ZKey()
    zk.header = GrothHeader("bn128", Snarkjs, nvars, npubs, domsiz, log2siz)   # parsed keys are Snarkjs (zkey.nim:129)

    def take(k):
        nonlocal pos
        b = bytes(s2[pos:pos + k])
        pos += k
        return b
    a1, b1, b2, g2, d1, d2 = take(64), take(64), take(128), take(128), take(64), take(128)
    zk.specPoints = SpecPoints(alpha1=a1, beta1=b1, beta2=b2, gamma2=g2, delta1=d1, delta2=d2)

    def pts(i, psz, n):
        s = one(i)
        assert len(s) == psz * n, "unexpected section length"
        return s.tobytes()
    zk.pointsIC = pts(3, 64, npubs + 1)                                  # zkey.nim:196-199
    zk.pPoints = ProverPoints(pointsA1=pts(5, 64, nvars), pointsB1=pts(6, 64, nvars), pointsB2=pts(7, 128, nvars),
                              pointsC1=pts(8, 64, nvars - npubs - 1), pointsH1=pts(9, 64, domsiz))
    s4 = one(4)                                                          # zkey.nim:169-192
    (ncoeffs,) = struct.unpack_from("<I", s4, 0)
    assert len(s4) == 4 + ncoeffs * (32 + 12), "unexpected section length"
    rec = np.frombuffer(s4, dtype=np.dtype([("m", "<u4"), ("r", "<u4"), ("c", "<u4"), ("v", "u1", (32,))]),
                        count=ncoeffs, offset=4)
    assert ncoeffs == 0 or int(rec["m"].max()) <= 2, "invalid matrix selector"
    assert ncoeffs == 0 or int(rec["r"].max()) < domsiz, "row index out of range"
    assert ncoeffs == 0 or int(rec["c"].max()) < nvars, "column index out of range"
    if rawCoeffs:
        zk.coeffs, zk.coeffsSection4 = [], bytes(s4)
        if check:
            checkZKeyPoints(zk, ctx)
        return zk
    raw = rec["v"].tobytes()
    # file integer = c * R^2 ; in-memory Montgomery limbs of c = c * R = file integer * R^-1   (unmarshalFrWTF)
    ms, rs, cs = rec["m"].tolist(), rec["r"].tolist(), rec["c"].tolist()
    coeffs = []
    for i in range(ncoeffs):
        v = int.from_bytes(raw[32 * i:32 * i + 32], "little")
        coeffs.append((ms[i], rs[i], cs[i], (v * F.frInvMontR % _R).to_bytes(32, "little")))
    zk.coeffs = coeffs
    if check:
        checkZKeyPoints(zk, ctx)
    return zk


def writeZKey(fname: str, zk: ZKey) -> None:
    """Inverse of parseZKey (sections 1-9 of zkey.nim:6-91; section 10, the ceremony contributions, is not
    needed for proving and is omitted)."""
    h, sp, pp = zk.header, zk.specPoints, zk.pPoints
    le32 = lambda x: int(x).to_bytes(32, "little")                       # noqa: E731
    s2 = (struct.pack("<I", 32) + le32(F.primeP) + struct.pack("<I", 32) + le32(F.primeR) +
          struct.pack("<III", h.nvars, h.npubs, h.domainSize) +
          sp.alpha1 + sp.beta1 + sp.beta2 + sp.gamma2 + sp.delta1 + sp.delta2)
    s4 = bytearray(struct.pack("<I", len(zk.coeffs)))
    for (m, r, c, v) in zk.coeffs:
        x = int.from_bytes(v, "little") * F.frMontR % _R                 # c*R -> c*R^2
        s4 += struct.pack("<III", m, r, c) + x.to_bytes(32, "little")
    writeContainer("zkey", 1, fname, [(1, struct.pack("<I", 1)), (2, s2), (3, zk.pointsIC), (4, bytes(s4)),
                                      (5, pp.pointsA1), (6, pp.pointsB1), (7, pp.pointsB2), (8, pp.pointsC1),
                                      (9, pp.pointsH1)])
