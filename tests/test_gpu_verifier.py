"""GPU parity for the verifier row through the C ABI: the raw ate pairing bit-exact against the
oracle, and verifyProof on the reference's toy circuit (tests/groth16/testProver.nim:59-73: prove -> verify),
incl. the negative cases (tampered proof / public input, malformed points, subgroup check)."""
import pytest

from oracle import bn254_ref as o
from tests import inputs as I

pytestmark = pytest.mark.gpu


def _gt(raw):
    """384-byte flat Fp12 (6 x Fp2 over w^k, Montgomery) -> the oracle's degree-12 polynomial basis"""
    out = [0] * 12
    for k in range(6):
        a = o.fp_from_mont_bytes(raw[64 * k:64 * k + 32])
        b = o.fp_from_mont_bytes(raw[64 * k + 32:64 * k + 64])
        out = [(x + y) % o.P for x, y in zip(out, o._emb((a, b), k))]
    return out


def test_pairing_vs_oracle_and_bilinear(ctx):
    rng = o.SplitMix64(31)
    ks = [(rng.fr(), rng.fr()) for _ in range(3)]
    Ps = [o.G1.mul(a, o.GEN1) for a, _ in ks] + [o.INF_G1, o.GEN1, o.G1.mul(ks[0][0] * ks[0][1] % o.R, o.GEN1)]
    Qs = [o.G2.mul(b, o.GEN2) for _, b in ks] + [o.GEN2, o.INF_G2, o.GEN2]
    raw = ctx.pairing(b"".join(o.g1_to_bytes(p) for p in Ps), b"".join(o.g2_to_bytes(q) for q in Qs))
    gts = [_gt(raw[384 * i:384 * i + 384]) for i in range(len(Ps))]
    assert gts[0] == o.pairing(Ps[0], Qs[0])                 # bit-exact incl. the final exponentiation
    assert gts[3] == o._f12_one() and gts[4] == o._f12_one()  # infinity on either side
    assert gts[5] == gts[0]                                   # e(aP, bQ) == e(abP, Q)
    assert gts[1] != gts[2] and gts[1] != o._f12_one()
    assert ctx.pairing(b"", b"") == b""


def _toy(ctx, flavour=1):
    from nim_groth16_amd import Mask, Witness, generateProofWithMask
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    rng = o.SplitMix64(5)
    a, b, g, d, t = (rng.fr() for _ in range(5))
    zk = fakeCircuitSetup(R1CS(8, 1, 1, 3, o.toy_r1cs().constraints), ToxicWaste(a, b, g, d, t), flavour, ctx)
    wt = Witness("bn128", 8, I.fr_mont_bytes(o.TOY_WITNESS))
    m = o.SplitMix64(6)
    return zk, [generateProofWithMask(0, False, zk, wt, Mask(m.fr(), m.fr()), ctx) for _ in range(2)]


@pytest.mark.parametrize("flavour", [0, 1])
def test_verify_toy_proofs(ctx, flavour):
    import dataclasses
    from nim_groth16_amd import extractVKey, loadVerifyingKey, verifyProof, verifyProofs
    zk, proofs = _toy(ctx, flavour)
    vkey = extractVKey(zk)
    assert vkey.npubs == 2
    assert verifyProof(vkey, proofs[0], ctx)                  # testProver.nim:65-73
    dev = loadVerifyingKey(vkey, ctx)
    good, other = proofs
    bad_c = dataclasses.replace(good, pi_c=other.pi_c)
    bad_a = dataclasses.replace(good, pi_a=o.g1_to_bytes(o.G1.mul(7, o.GEN1)))
    bad_pub = dataclasses.replace(good, publicIO=I.fr_mont_bytes([1, 2024, 1022]))
    zero_pub = dataclasses.replace(good, publicIO=I.fr_mont_bytes([1, 0, 0]))
    res = verifyProofs(dev, [good, bad_c, other, bad_a, bad_pub, zero_pub], ctx, subgroup=True)
    assert res == [True, False, True, False, False, False]
    # agreement with the oracle's verifier on the same objects
    from tests.test_gpu_prover import _zkey_to_oracle
    oz = _zkey_to_oracle(zk)
    for prf, exp in zip([good, bad_c, bad_pub], [True, False, False]):
        ref = o.Proof(I.fr_from_mont(prf.publicIO), o.g1_from_bytes(prf.pi_a), o.g2_from_bytes(prf.pi_b),
                      o.g1_from_bytes(prf.pi_c))
        assert o.verify_proof(oz, ref) == exp


def test_verify_rejects_malformed_points(ctx):
    import dataclasses
    from nim_groth16_amd import extractVKey, loadVerifyingKey, verifyProof
    zk, (good, _) = _toy(ctx)
    dev = loadVerifyingKey(extractVKey(zk), ctx)
    off_g1 = o.fp_to_mont_bytes(5) + o.fp_to_mont_bytes(7)             # not on y^2 = x^3 + 3
    with pytest.raises(AssertionError, match="pi_a is not in G1"):
        verifyProof(dev, dataclasses.replace(good, pi_a=off_g1), ctx)
    with pytest.raises(AssertionError, match="pi_c is not in G1"):
        verifyProof(dev, dataclasses.replace(good, pi_c=off_g1), ctx)
    with pytest.raises(AssertionError, match="pi_b is not in G2"):
        verifyProof(dev, dataclasses.replace(good, pi_b=bytes(good.pi_b[:64]) + bytes(64)), ctx)
    # a point of the twist outside the order-r subgroup: on the curve (the reference would accept it as input
    # and fail the pairing equation); with subgroup=True it is refused up front
    x = (3, 1)
    while True:
        rhs = o.fp2_add(o.fp2_mul(o.fp2_sqr(x), x), o.TWIST_B)
        y = _fp2_sqrt(rhs)
        if y is not None and not o.G2.is_inf(o.G2.mul(o.R, (x, y))):
            break
        x = (x[0] + 1, x[1])
    rogue = dataclasses.replace(good, pi_b=o.g2_to_bytes((x, y)))
    assert dev.verify([(rogue.pi_a, rogue.pi_b, rogue.pi_c)], rogue.publicIO) == [0]
    assert dev.verify([(rogue.pi_a, rogue.pi_b, rogue.pi_c)], rogue.publicIO, subgroup=True) == [-4]
    assert dev.verify([], b"") == []


def _fp2_sqrt(a):
    """square root in Fp2 = Fp[u]/(u^2+1) (p = 3 mod 4), or None"""
    p = o.P
    if a == (0, 0):
        return (0, 0)
    norm = (a[0] * a[0] + a[1] * a[1]) % p
    s = pow(norm, (p + 1) // 4, p)
    if s * s % p != norm:
        return None
    for sgn in (s, p - s):
        t = (a[0] + sgn) * pow(2, -1, p) % p
        x0 = pow(t, (p + 1) // 4, p)
        if x0 * x0 % p == t and x0:
            x1 = a[1] * pow(2 * x0, -1, p) % p
            if o.fp2_sqr((x0, x1)) == (a[0] % p, a[1] % p):
                return (x0, x1)
    return None
