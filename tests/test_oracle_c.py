"""Pins the C oracle (oracle/g16_oracle.c) against the pure-Python oracle on small and mid sizes."""
import random

import pytest

from oracle import bn254_ref as o
from tests import inputs as I


def test_field_ops(orc):
    rng = random.Random(1)
    for field, mod in ((0, o.P), (1, o.R)):
        Rm = o.MONT % mod
        Ri = pow(Rm, -1, mod)
        for _ in range(100):
            a, b = rng.randrange(mod), rng.randrange(mod)
            f = lambda op, x, y=0: int.from_bytes(orc.field_op(field, op, x.to_bytes(32, "little"), y.to_bytes(32, "little")), "little")   # noqa: E731
            assert f(0, a, b) == (a + b) % mod and f(1, a, b) == (a - b) % mod
            assert f(2, a, b) == a * b * Ri % mod
            assert f(6, a) == a * pow(2, -1, mod) % mod
            assert f(8, a) == a * Ri % mod and f(9, a) == a * Rm % mod
        a = rng.randrange(1, mod)
        assert f(7, a * Rm % mod) == pow(a, -1, mod) * Rm % mod


@pytest.mark.parametrize("group", [1, 2])
def test_fixed_base_and_msm(orc, group):
    C, gen = (o.G1, o.GEN1) if group == 1 else (o.G2, o.GEN2)
    dec = o.g1_from_bytes if group == 1 else o.g2_from_bytes
    psz = 64 if group == 1 else 128
    ks = I.uniform_scalars(12, 7) + [0, 1, o.R - 1]
    pts = orc.fixed_base(group, I.fr_mont_bytes(ks))
    for i, k in enumerate(ks):
        assert dec(pts[psz * i:psz * (i + 1)]) == C.mul(k, gen)
    cs = I.uniform_scalars(len(ks), 8)
    exp = C.mul(sum(c * k for c, k in zip(cs, ks)) % o.R, gen)
    assert dec(orc.msm_naive(group, I.fr_mont_bytes(cs), pts)) == exp
    assert dec(orc.msm(group, I.fr_mont_bytes(cs), pts)) == exp
    assert dec(orc.msm(group, I.fr_std_bytes(cs), pts, mont=False)) == exp
    assert dec(orc.msm(group, b"", b"")) == C.inf


def test_msm_multithreaded_chunks_match_closed_form(orc):
    n = 3000                                  # > 128 * threads: exercises the chunking rule of msm.nim:98-119
    ks, pts = I.points_with_logs(orc, 1, n, seed=5)
    cs = I.circom_like_scalars(n, 6)
    for threads in (1, 3, 0):
        assert orc.msm(1, I.fr_mont_bytes(cs), pts, threads=threads) == I.expected_from_logs(1, cs, ks)


@pytest.mark.parametrize("lg", [0, 1, 2, 3, 7])
def test_ntt_and_quotient(orc, lg):
    n = 1 << lg
    xs = I.uniform_scalars(n, 30 + lg)
    D = o.Domain(n)
    assert I.fr_from_mont(orc.ntt(I.fr_mont_bytes(xs), lg)) == o.forward_ntt(xs, D)
    assert I.fr_from_mont(orc.ntt(I.fr_mont_bytes(xs), lg, inverse=True)) == o.inverse_ntt(xs, D)
    if lg >= 1:
        A, B, Cc = (I.uniform_scalars(n, s) for s in (1, 2, 3))
        q = orc.quotient_snarkjs(*(I.fr_mont_bytes(v) for v in (A, B, Cc)), lg)
        assert I.fr_from_mont(q) == o.compute_snarkjs_scalar_coeffs(A, B, Cc)
        # JensGroth flavour (computeQuotientPointwise, prover.nim:118-148), serial and 3-task
        for par in (False, True):
            qj = orc.quotient_jensgroth(*(I.fr_mont_bytes(v) for v in (A, B, Cc)), lg, parallel=par)
            assert I.fr_from_mont(qj) == o.compute_quotient_pointwise(A, B, Cc)


def test_quotient_jensgroth_is_the_quotient_polynomial(orc):
    """size-independent pin of the C JensGroth quotient at a mid size: for Az, Bz, Cz = Az*Bz on the domain
    (what buildABC produces, prover.nim:56-73) the output q satisfies q(x) * (x^n - 1) == A(x)*B(x) - C(x) at
    random points x, with A, B, C the interpolants (the defining property of prover.nim:116-117)."""
    lg, n = 10, 1 << 10
    A, B = I.uniform_scalars(n, 51), I.uniform_scalars(n, 52)
    Cc = [a * b % o.R for a, b in zip(A, B)]
    q = I.fr_from_mont(orc.quotient_jensgroth(*(I.fr_mont_bytes(v) for v in (A, B, Cc)), lg))
    ca, cb, cc = (I.fr_from_mont(orc.ntt(I.fr_mont_bytes(v), lg, inverse=True)) for v in (A, B, Cc))
    ev = lambda cs, x: sum(c * pow(x, i, o.R) for i, c in enumerate(cs)) % o.R     # noqa: E731
    for x in I.uniform_scalars(3, 53):
        assert ev(q, x) * (pow(x, n, o.R) - 1) % o.R == (ev(ca, x) * ev(cb, x) - ev(cc, x)) % o.R


def test_domain_generator(orc):
    import ctypes
    out = ctypes.create_string_buffer(32)
    for lg in (1, 10, 20, 28):
        orc.lib.orc_domain_gen(lg, out)
        assert o.fr_from_mont_bytes(out.raw) == o.Domain(1 << lg).domainGen
