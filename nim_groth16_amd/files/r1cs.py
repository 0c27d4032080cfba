"""groth16/files/r1cs.nim: `.r1cs` v1 -- header, constraints (three sparse linear combinations each),
wire-to-label map.  Coefficients are in standard form.  Only the fake setup needs it."""
from __future__ import annotations

import struct

from .. import bn128 as F
from ..fake_setup import R1CS
from .container import parseContainer, parsePrimeField, writeContainer


def parseR1CS(fname: str) -> R1CS:
    """r1cs.nim:170-174"""
    sec = parseContainer("r1cs", 1, fname)
    s1 = sec[1][0]                                                       # r1cs.nim:84-108
    n8r, r, pos = parsePrimeField(s1, 0)
    assert len(s1) == 4 + n8r + 16 + 8 + 4, "unexpected section length"
    assert r == F.primeR, "expecting the alt-bn128 curve"
    nWires, nPubOut, nPubIn, nPrivIn, nLabels, nConstr = struct.unpack_from("<IIIIQI", s1, pos)
    s2 = sec[2][0]                                                       # r1cs.nim:112-142
    pos = 0

    def lincomb():
        nonlocal pos
        (nterms,) = struct.unpack_from("<I", s2, pos)
        pos += 4
        terms = []
        for _ in range(nterms):
            (idx,) = struct.unpack_from("<I", s2, pos)
            terms.append((idx, int.from_bytes(bytes(s2[pos + 4:pos + 36]), "little")))
            pos += 36
        return terms
    cons = [(lincomb(), lincomb(), lincomb()) for _ in range(nConstr)]
    r1 = R1CS(nWires, nPubOut, nPubIn, nPrivIn, cons)
    if 3 in sec:                                                         # r1cs.nim:146-152
        s3 = sec[3][0]
        assert len(s3) == 8 * nWires, "unexpected section length"
        r1.wireToLabel = list(struct.unpack_from(f"<{nWires}Q", s3, 0))
    return r1


def writeR1CS(fname: str, r1cs: R1CS) -> None:
    s1 = (struct.pack("<I", 32) + F.primeR.to_bytes(32, "little") +
          struct.pack("<IIIIQI", r1cs.nWires, r1cs.nPubOut, r1cs.nPubIn, r1cs.nPrivIn, 0, len(r1cs.constraints)))
    s2 = bytearray()
    for abc in r1cs.constraints:
        for lc in abc:
            s2 += struct.pack("<I", len(lc))
            for (idx, v) in lc:
                s2 += struct.pack("<I", idx) + (int(v) % F.primeR).to_bytes(32, "little")
    s3 = struct.pack(f"<{r1cs.nWires}Q", *range(r1cs.nWires))
    writeContainer("r1cs", 1, fname, [(1, s1), (2, bytes(s2)), (3, s3)])
