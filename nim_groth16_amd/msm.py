"""Host mirror of groth16/bn128/msm.nim (reference) over the C ABI.

The reference's procs take `seq[Fr]` / `seq[G1]`; here a "seq" is a bytes-like object holding the
elements in the reference's in-memory layout (32-byte Montgomery Fr, 64-byte G1, 128-byte G2;
include/g16hip.h), so a test can hand the very same bytes to the oracle.
"""
from __future__ import annotations

from ._lib import default_context


def _check_lengths(coeffs, points, psz):
    # msm.nim:97 / :131  assert( N == points.len, "incompatible sequence lengths" )
    assert len(coeffs) % 32 == 0 and len(points) % psz == 0, "malformed sequence"
    n = len(coeffs) // 32
    assert n == len(points) // psz, "incompatible sequence lengths"
    return n


def msmMultiThreadedG1(nthreads_hint: int, coeffs: bytes, points: bytes, ctx=None) -> bytes:
    """msm.nim:89-124.  `nthreads_hint` is accepted for signature parity and ignored: the chunking over
    host threads (msm.nim:98-119) is replaced by the GPU's own decomposition; the result -- one canonical
    affine G1 point, (0,0) for infinity -- is the same for every chunking."""
    n = _check_lengths(coeffs, points, 64)
    return (ctx or default_context()).msm(1, coeffs, points, n, mont=True)


def msmMultiThreadedG2(nthreads_hint: int, coeffs: bytes, points: bytes, ctx=None) -> bytes:
    """msm.nim:128-158."""
    n = _check_lengths(coeffs, points, 128)
    return (ctx or default_context()).msm(2, coeffs, points, n, mont=True)


def msmG1(coeffs: bytes, points: bytes, ctx=None) -> bytes:
    """msm.nim:202  (= msmConstantineG1, msm.nim:35-59)"""
    return msmMultiThreadedG1(0, coeffs, points, ctx)


def msmG2(coeffs: bytes, points: bytes, ctx=None) -> bytes:
    """msm.nim:203  (= msmConstantineG2, msm.nim:63-83)"""
    return msmMultiThreadedG2(0, coeffs, points, ctx)
