"""Host mirror of groth16/prover.nim: generateProof / generateProofWithMask / generateProofWithTrivialMask.
The whole proof (buildABC, quotient NTTs, five MSMs) is one g16_prove call on the GPU; see
nim_groth16_amd/csrc/prover.hip for the launch sequence."""
from __future__ import annotations

import ctypes
import secrets
from dataclasses import dataclass
from typing import Optional

from . import bn128 as F
from ._lib import PkeyDesc, ProvingKey, default_context
from .zkey_types import JensGroth, Snarkjs, ZKey, packCoeffs


@dataclass
class Witness:                       # files/witness.nim:27-32
    curve: str
    nvars: int
    values: bytes                    # nvars Fr: Montgomery (the Nim seq[Fr] layout), or ...
    std: bool = False                # ... standard form (raw .wtns bytes, files/witness.nim:14) when True


@dataclass
class Mask:                          # prover.nim:210-213
    r: int
    s: int


@dataclass
class Proof:                         # prover.nim:37-43
    publicIO: bytes                  # (npubs+1) Fr, Montgomery, incl. the leading constant 1 (prover.nim:238-240)
    pi_a: bytes
    pi_b: bytes
    pi_c: bytes
    curve: str = "bn128"


def _cbuf(b: bytes):
    return ctypes.create_string_buffer(b, len(b)) if b else ctypes.create_string_buffer(1)


def loadProvingKey(zkey: ZKey, ctx=None, shard_index: int = 0, shard_count: int = 1) -> ProvingKey:
    """Uploads the ZKey once (the reference parses it once per run, files/zkey.nim:241-245).  With
    shard_count > 1 only this rank's contiguous index range of every point set is kept (msm.nim:105-115)."""
    ctx = ctx or default_context()
    hdr, pts, spec = zkey.header, zkey.pPoints, zkey.specPoints
    # shape asserts of generateProofWithMask (prover.nim:270-276)
    assert len(pts.pointsA1) == 64 * hdr.nvars and len(pts.pointsB1) == 64 * hdr.nvars
    assert len(pts.pointsB2) == 128 * hdr.nvars
    assert len(pts.pointsH1) == 64 * hdr.domainSize
    assert len(pts.pointsC1) == 64 * (hdr.nvars - hdr.npubs - 1)
    raw4 = getattr(zkey, "coeffsSection4", None)      # the .zkey's coefficient section, unparsed (parseZKey rawCoeffs)
    bufs = [_cbuf(x) for x in (pts.pointsA1, pts.pointsB1, pts.pointsB2, pts.pointsC1, pts.pointsH1,
                               b"" if raw4 is not None else packCoeffs(zkey.coeffs), spec.alpha1, spec.beta1,
                               spec.delta1, spec.beta2, spec.delta2)]
    addr = [ctypes.cast(b, ctypes.c_void_p) for b in bufs]
    desc = PkeyDesc(hdr.nvars, hdr.npubs, hdr.logDomainSize, hdr.flavour, addr[0], addr[1], addr[2], addr[3],
                    addr[4], None if raw4 is not None else addr[5], 0 if raw4 is not None else len(zkey.coeffs),
                    addr[6], addr[7], addr[8], addr[9], addr[10], shard_index, shard_count)
    return ProvingKey(ctx, desc, bufs, section4=raw4)


def loadGroupKey(zkey: ZKey, group):
    """The ZKey sharded over the members of a _lib.DeviceGroup (g16_group_pkey_create): member g keeps index range g of
    every ProverPoints array (msm.nim:105-115)."""
    hdr, pts, spec = zkey.header, zkey.pPoints, zkey.specPoints
    assert len(pts.pointsA1) == 64 * hdr.nvars and len(pts.pointsB1) == 64 * hdr.nvars
    assert len(pts.pointsB2) == 128 * hdr.nvars and len(pts.pointsH1) == 64 * hdr.domainSize
    assert len(pts.pointsC1) == 64 * (hdr.nvars - hdr.npubs - 1)
    bufs = [_cbuf(x) for x in (pts.pointsA1, pts.pointsB1, pts.pointsB2, pts.pointsC1, pts.pointsH1,
                               packCoeffs(zkey.coeffs), spec.alpha1, spec.beta1, spec.delta1, spec.beta2,
                               spec.delta2)]
    addr = [ctypes.cast(b, ctypes.c_void_p) for b in bufs]
    desc = PkeyDesc(hdr.nvars, hdr.npubs, hdr.logDomainSize, hdr.flavour, addr[0], addr[1], addr[2], addr[3],
                    addr[4], addr[5], len(zkey.coeffs), addr[6], addr[7], addr[8], addr[9], addr[10], 0, 1)
    return group.load_key(desc, bufs)


_pkey_cache = {}


def _pkey_for(zkey: ZKey, ctx) -> ProvingKey:
    key = (id(zkey), id(ctx))
    if key not in _pkey_cache:
        _pkey_cache[key] = (zkey, loadProvingKey(zkey, ctx))
    return _pkey_cache[key][1]


@dataclass
class ABC:                           # prover.nim:49-53
    valuesAz: bytes
    valuesBz: bytes
    valuesCz: bytes


def buildABC(zkey: ZKey, witness: bytes, ctx=None, pkey: Optional[ProvingKey] = None) -> ABC:
    """prover.nim:56-73 -- A*z, B*z and their pointwise product on the domain (sparse mat-vec on the GPU);
    witness = nvars Fr in the Nim seq[Fr] layout"""
    ctx = ctx or default_context()
    assert zkey.header.nvars * 32 == len(witness), "wrong witness length"
    a, b, c = (pkey or _pkey_for(zkey, ctx)).build_abc(witness)
    return ABC(a, b, c)


def _log2n(abc: ABC) -> int:
    n = len(abc.valuesAz) // 32
    assert len(abc.valuesBz) == len(abc.valuesCz) == 32 * n and n & (n - 1) == 0 and n > 0   # prover.nim:160-161
    return n.bit_length() - 1


def computeSnarkjsScalarCoeffs(nthreads: int, abc: ABC, ctx=None) -> bytes:
    """prover.nim:158-181 -- the H-MSM scalars of a snarkjs-flavour key (6 NTTs + pointwise, on the GPU)"""
    return (ctx or default_context()).quotient(abc.valuesAz, abc.valuesBz, abc.valuesCz, _log2n(abc), Snarkjs)


def computeQuotientPointwise(nthreads: int, abc: ABC, ctx=None) -> bytes:
    """prover.nim:118-148 -- coefficients of the quotient polynomial (JensGroth flavour, 7 NTTs)"""
    return (ctx or default_context()).quotient(abc.valuesAz, abc.valuesBz, abc.valuesCz, _log2n(abc), JensGroth)


def generateProofWithMask(nthreads: int, printTimings: bool, zkey: ZKey, wtns: Witness, mask: Mask,
                          ctx=None, pkey: Optional[ProvingKey] = None) -> Proof:
    """prover.nim:215-304.  `nthreads` / `printTimings` are accepted for signature parity (the work is on
    the GPU; use Context.profile for timings)."""
    ctx = ctx or default_context()
    assert zkey.header.curve == wtns.curve                       # prover.nim:224
    assert zkey.header.nvars * 32 == len(wtns.values), "wrong witness length"   # prover.nim:236
    pkey = pkey or _pkey_for(zkey, ctx)
    r = F.frToMontBytes(mask.r) if mask.r % F.primeR else None
    s = F.frToMontBytes(mask.s) if mask.s % F.primeR else None
    pi_a, pi_b, pi_c = pkey.prove(wtns.values, mont=not wtns.std, r=r, s=s)
    pubIO = wtns.values[: 32 * (zkey.header.npubs + 1)]
    if wtns.std:                     # Proof.publicIO is seq[Fr]: Montgomery in memory
        pubIO = F.frSeqToMontBytes(int.from_bytes(pubIO[i:i + 32], "little") for i in range(0, len(pubIO), 32))
    return Proof(pubIO, pi_a, pi_b, pi_c)


def generateProofWithTrivialMask(nthreads: int, printTimings: bool, zkey: ZKey, wtns: Witness, ctx=None) -> Proof:
    """prover.nim:308-310"""
    return generateProofWithMask(nthreads, printTimings, zkey, wtns, Mask(0, 0), ctx)


def generateProof(nthreads: int, printTimings: bool, zkey: ZKey, wtns: Witness, ctx=None) -> Proof:
    """prover.nim:312-319 (the reference draws r, s from a clock-seeded PRNG, rnd.nim:15-24; here from the OS)."""
    mask = Mask(secrets.randbelow(F.primeR), secrets.randbelow(F.primeR))
    return generateProofWithMask(nthreads, printTimings, zkey, wtns, mask, ctx)
