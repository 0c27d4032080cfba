"""Host mirror of groth16/math/domain.nim + groth16/math/ntt.nim (reference) over the C ABI."""
from __future__ import annotations

from ._lib import default_context


class Domain:
    """math/domain.nim:15-21.  Only the size fields live on the host; the generator
    (gen28^(2^(28-log2 n)), domain.nim:31-33) is derived on the device."""

    def __init__(self, domainSize: int, logDomainSize: int):
        self.domainSize = domainSize
        self.logDomainSize = logDomainSize


def createDomain(size: int) -> Domain:
    """math/domain.nim:28-46"""
    log2 = (size - 1).bit_length() if size > 0 else -1
    assert size > 0 and (1 << log2) == size, "domain must have a power-of-two size"
    assert log2 <= 28, "domain larger than the 2-adicity of r-1"
    return Domain(size, log2)


def _check(src: bytes, D: Domain):
    # ntt.nim:56-57 / :140-141
    assert D.domainSize == (1 << D.logDomainSize), "domain must have a power-of-two size"
    assert D.domainSize * 32 == len(src), "input must have the same size as the domain"


def forwardNTT(src: bytes, D: Domain, ctx=None) -> bytes:
    """math/ntt.nim:55-77: y_k = sum_i x_i g^(ik), natural order, unscaled."""
    _check(src, D)
    return (ctx or default_context()).ntt(src, D.logDomainSize, inverse=False)


def inverseNTT(src: bytes, D: Domain, ctx=None) -> bytes:
    """math/ntt.nim:139-161: exact inverse of forwardNTT (1/n included)."""
    _check(src, D)
    return (ctx or default_context()).ntt(src, D.logDomainSize, inverse=True)


def extendAndForwardNTT(src: bytes, D: Domain, ctx=None) -> bytes:
    """math/ntt.nim:81-91: zero-pad to the domain size, then forwardNTT."""
    n = len(src) // 32
    assert n <= D.domainSize
    return forwardNTT(src + b"\x00" * (32 * (D.domainSize - n)), D, ctx)


def polyForwardNTT(coeffs: bytes, D: Domain, ctx=None) -> bytes:
    """poly.nim:255-260 -- a Poly's coefficients (seq[Fr] bytes) evaluated on the domain.  The reference asserts
    `coeffs.len <= domainSize` and then calls forwardNTT, which needs equality; here shorter polynomials are padded,
    which is what the assert intends."""
    assert len(coeffs) // 32 <= D.domainSize, "the input polynomial must have at most `domainSize` coefficients"
    return extendAndForwardNTT(coeffs, D, ctx)


def polyInverseNTT(ys: bytes, D: Domain, ctx=None) -> bytes:
    """poly.nim:264-268 -- values on the domain -> coefficients"""
    return inverseNTT(ys, D, ctx)
