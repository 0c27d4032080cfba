"""Pins the oracle with the reference's own fixture: toy R1CS + witness -> fake setup -> prove -> verify, both
flavours (tests/groth16/testProver.nim:17-73), plus the derived values recorded in SURVEY.md 8c."""
import pytest

from oracle import bn254_ref as o


def test_toy_abc_and_snarkjs_scalars():
    cf = o.r1cs_to_coeffs(o.toy_r1cs())
    Az, Bz, Cz = o.build_abc(cf, 8, o.TOY_WITNESS)
    assert Az == [0, 7, 13, 1, 2023, 1022, 0, 0]
    assert Bz == [0, 11, 77, 0, 0, 0, 0, 0]
    assert Cz == [0, 77, 1001, 0, 0, 0, 0, 0]
    qs = o.compute_snarkjs_scalar_coeffs(Az, Bz, Cz)
    assert qs[0] == 0x28e8f3caa9108d0537c50b93a0484ea5c1dd002eeae0122b5f73f5761a6b8c9b
    assert qs[7] == 0x0206f9301fbd1f65c9a7221c8a0bfd2febf0ece260a7bfa2bd04e19b41c84abf
    # independent check: qs[j] = (A*B - C)(eta * w^j) by direct polynomial evaluation
    D = o.Domain(8)
    eta = o.Domain(16).domainGen
    pa, pb, pc = (o.inverse_ntt(v, D) for v in (Az, Bz, Cz))
    ev = lambda p, x: sum(c * pow(x, i, o.R) for i, c in enumerate(p)) % o.R    # noqa: E731
    for j in (0, 3, 7):
        x = eta * pow(D.domainGen, j, o.R) % o.R
        assert qs[j] == (ev(pa, x) * ev(pb, x) - ev(pc, x)) % o.R


@pytest.mark.parametrize("flavour", [o.JENS_GROTH, o.SNARKJS])
def test_toy_prove_and_verify(flavour):
    rng = o.SplitMix64(99)
    tox = o.ToxicWaste(*[rng.fr() for _ in range(5)])
    zk = o.fake_circuit_setup(o.toy_r1cs(), tox, flavour)
    for (r, s) in ((0, 0), (rng.fr(), rng.fr())):
        pr = o.generate_proof_with_mask(zk, o.TOY_WITNESS, r, s)
        assert o.verify_proof(zk, pr)
        assert pr.publicIO == [1, 2023, 1022]
        bad = o.Proof(pr.publicIO, pr.pi_a, pr.pi_b, o.G1.add(pr.pi_c, o.GEN1))
        assert not o.verify_proof(zk, bad)
        bad2 = o.Proof([1, 2024, 1022], pr.pi_a, pr.pi_b, pr.pi_c)
        assert not o.verify_proof(zk, bad2)


def test_pairing_is_bilinear_and_nondegenerate():
    e0 = o.pairing(o.GEN1, o.GEN2)
    assert e0 != o._f12_one()
    assert o.pairing(o.G1.mul(5, o.GEN1), o.G2.mul(7, o.GEN2)) == o._f12_pow(e0, 35)
    assert o._f12_pow(e0, o.R) == o._f12_one()


def test_ntt_conventions():
    rng = o.SplitMix64(1)
    for lg in (0, 1, 2, 4, 6):
        n = 1 << lg
        D = o.Domain(n)
        xs = [rng.fr() for _ in range(n)]
        assert o.forward_ntt(xs, D) == o.naive_dft(xs, D)           # natural order, unscaled
        assert o.inverse_ntt(o.forward_ntt(xs, D), D) == xs         # inverse includes 1/n
    D = o.Domain(8)
    assert o.extend_and_forward_ntt([1, 2, 3], D) == o.forward_ntt([1, 2, 3, 0, 0, 0, 0, 0], D)


def test_msm_definitions_agree():
    rng = o.SplitMix64(3)
    for C, gen in ((o.G1, o.GEN1), (o.G2, o.GEN2)):
        ks = [rng.fr() for _ in range(12)]
        pts = [C.mul(k, gen) for k in ks] + [C.inf]
        cs = [rng.fr() for _ in range(12)] + [5]
        exp = C.mul(sum(c * k for c, k in zip(cs, ks)) % o.R, gen)
        assert C.msm_naive(cs, pts) == exp == C.msm_pippenger(cs, pts)
        assert o.msm_multithreaded(C, 4, cs, pts) == exp
        assert C.msm_naive([], []) == C.inf
