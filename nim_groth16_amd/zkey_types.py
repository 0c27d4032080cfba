"""Host mirror of groth16/zkey_types.nim: the in-memory proving key.  Field elements and points are kept
as `bytes` in the reference's memory layout (Montgomery, little-endian; include/g16hip.h)."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import List, Tuple

JensGroth, Snarkjs = 0, 1            # Flavour, zkey_types.nim:10-12
MatrixA, MatrixB, MatrixC = 0, 1, 2  # MatrixSel, zkey_types.nim:43-46


@dataclass
class GrothHeader:                   # zkey_types.nim:14-22
    curve: str = "bn128"
    flavour: int = Snarkjs
    nvars: int = 0
    npubs: int = 0
    domainSize: int = 0
    logDomainSize: int = 0


@dataclass
class SpecPoints:                    # zkey_types.nim:24-31 (alphaBeta is a verifier-only pairing value)
    alpha1: bytes = bytes(64)
    beta1: bytes = bytes(64)
    beta2: bytes = bytes(128)
    gamma2: bytes = bytes(128)
    delta1: bytes = bytes(64)
    delta2: bytes = bytes(128)


@dataclass
class ProverPoints:                  # zkey_types.nim:36-41
    pointsA1: bytes = b""
    pointsB1: bytes = b""
    pointsB2: bytes = b""
    pointsC1: bytes = b""
    pointsH1: bytes = b""


@dataclass
class ZKey:                          # zkey_types.nim:54-59
    header: GrothHeader = field(default_factory=GrothHeader)
    specPoints: SpecPoints = field(default_factory=SpecPoints)
    pointsIC: bytes = b""            # VerifierPoints, zkey_types.nim:33-34
    pPoints: ProverPoints = field(default_factory=ProverPoints)
    coeffs: List[Tuple[int, int, int, bytes]] = field(default_factory=list)   # (matrix, row, col, Fr mont bytes)
    # alternatively the .zkey file's coefficient section as it lies on disk (u32 count + 44-byte entries, values in the
    # double-Montgomery form of io.nim:134-139): parseZKey(rawCoeffs=True) keeps it, loadProvingKey hands it to
    # g16_pkey_create_zkey unparsed, and `coeffs` stays empty
    coeffsSection4: bytes = None


class _Coeff(ctypes.Structure):      # g16_coeff (include/g16hip.h)
    _fields_ = [("matrix", ctypes.c_uint32), ("row", ctypes.c_uint32), ("col", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32), ("value", ctypes.c_uint8 * 32)]


COEFF_DTYPE = [("m", "<u4"), ("r", "<u4"), ("c", "<u4"), ("x", "<u4"), ("v", "u1", (32,))]   # g16_coeff


class CoeffArray:
    """ZKey.coeffs as ONE numpy record array in the g16_coeff layout (large circuits: 10^7 entries as Python tuples
    would take gigabytes).  Iterates as the same (matrix, row, col, Fr bytes) tuples as the list form."""

    def __init__(self, arr):
        self.arr = arr

    def __len__(self):
        return len(self.arr)

    def __iter__(self):
        for rec in self.arr:
            yield (int(rec["m"]), int(rec["r"]), int(rec["c"]), rec["v"].tobytes())

    def __getitem__(self, i):
        rec = self.arr[i]
        return (int(rec["m"]), int(rec["r"]), int(rec["c"]), rec["v"].tobytes())


def packCoeffs(coeffs) -> bytes:
    import numpy as np
    if isinstance(coeffs, CoeffArray):
        return coeffs.arr.tobytes()
    n = len(coeffs)
    arr = np.zeros(n, dtype=np.dtype(COEFF_DTYPE))
    if n:
        arr["m"] = [c[0] for c in coeffs]
        arr["r"] = [c[1] for c in coeffs]
        arr["c"] = [c[2] for c in coeffs]
        arr["v"] = np.frombuffer(b"".join(c[3] for c in coeffs), dtype=np.uint8).reshape(n, 32)
    return arr.tobytes()
