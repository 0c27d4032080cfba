"""groth16/files/witness.nim: `.wtns` v2 -- header (n8r, r, nvars) + nvars x 32-byte values in STANDARD form
(canonical little-endian, witness.nim:14).  The values are handed to the GPU as they are (G16_SCALARS_STD);
the reference converts each one to Montgomery form on load (loadValuesFrStd, bn128/io.nim:208-212)."""
from __future__ import annotations

import struct

from .. import bn128 as F
from ..prover import Witness
from .container import parseContainer, parsePrimeField, writeContainer


def parseWitness(fname: str) -> Witness:
    """witness.nim:71-75"""
    sec = parseContainer("wtns", 2, fname)
    s1 = sec[1][0]                                                       # witness.nim:36-52
    n8r, r, pos = parsePrimeField(s1, 0)
    assert len(s1) == 4 + n8r + 4, "unexpected section length"
    assert n8r == 32, "expecting 256 bit prime"
    assert r == F.primeR, "expecting the alt-bn128 curve"
    (nvars,) = struct.unpack_from("<I", s1, pos)
    s2 = sec[2][0]                                                       # witness.nim:57-60
    assert len(s2) == 32 * nvars
    return Witness("bn128", nvars, s2.tobytes(), std=True)


def writeWitness(fname: str, values) -> None:
    """values: iterable of ints (standard form)."""
    vals = [int(v) % F.primeR for v in values]
    s1 = struct.pack("<I", 32) + F.primeR.to_bytes(32, "little") + struct.pack("<I", len(vals))
    writeContainer("wtns", 2, fname, [(1, s1), (2, b"".join(v.to_bytes(32, "little") for v in vals))])
