"""The GPU path against the committed fixtures alone (tests/golden/): nothing from oracle/ is imported or run here,
so this file pins the product's bytes on any box that has the product and the fixtures."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = json.load(open(os.path.join(G, "reference_fixture.json")))
VEC = json.load(open(os.path.join(G, "oracle_vectors.json")))
R = int(REF["constants"]["primeR"], 16)
P = int(REF["constants"]["primeP"], 16)


def fr(x):
    return (x % R * (1 << 256) % R).to_bytes(32, "little")


def test_msm_ntt_quotient_fixtures(ctx):
    for group in (1, 2):
        v = VEC[f"msm_g{group}"]
        sc, pts = bytes.fromhex(v["scalars"]), bytes.fromhex(v["points"])
        assert ctx.msm(group, sc, pts, v["n"]).hex() == v["result"]
        h = ctx.register_points(group, pts, v["n"])
        assert ctx.msm_points(h, sc).hex() == v["result"]
        h.release()
    n = VEC["ntt16"]
    assert ctx.ntt(bytes.fromhex(n["input"]), 4, False).hex() == n["forward"]
    assert ctx.ntt(bytes.fromhex(n["input"]), 4, True).hex() == n["inverse"]
    abc = [b"".join(fr(x) for x in REF[k]) for k in ("Az", "Bz", "Cz")]
    assert ctx.quotient(abc[0], abc[1], abc[2], 3, 1).hex() == VEC["quotient_toy"]["snarkjs"]
    assert ctx.quotient(abc[0], abc[1], abc[2], 3, 0).hex() == VEC["quotient_toy"]["jensgroth"]
    qs = ctx.quotient(abc[0], abc[1], abc[2], 3, 1)
    for k, val in REF["snarkjs_qs"].items():          # the reference-derived anchors
        assert qs[32 * int(k):32 * int(k) + 32] == fr(int(val, 16))


@pytest.mark.parametrize("flavour,name", [(1, "snarkjs"), (0, "jensgroth")])
def test_toy_proof_fixtures(ctx, flavour, name):
    from nim_groth16_amd import Mask, Witness, extractVKey, generateProofWithMask, loadProvingKey, verifyProof
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    t = REF["toy_circuit"]
    cons = [tuple([(w, int(v) % R) for w, v in part] for part in con) for con in t["constraints"]]
    tw = ToxicWaste(*(int(VEC["toxic_waste"][k], 16) for k in ("alpha", "beta", "gamma", "delta", "tau")))
    zk = fakeCircuitSetup(R1CS(t["nWires"], t["nPubOut"], t["nPubIn"], t["nPrivIn"], cons), tw, flavour, ctx)
    pk = loadProvingKey(zk, ctx)
    wt = Witness("bn128", t["nWires"], b"".join(fr(x) for x in t["witness"]))
    Az, Bz, Cz = pk.build_abc(wt.values)
    assert (Az, Bz, Cz) == tuple(b"".join(fr(x) for x in REF[k]) for k in ("Az", "Bz", "Cz"))   # prover.nim:56-73
    mask = Mask(int(VEC["mask"]["r"], 16), int(VEC["mask"]["s"], 16))
    for pr, tag in ((generateProofWithMask(0, False, zk, wt, Mask(0, 0), ctx, pkey=pk), "trivial_mask"),
                    (generateProofWithMask(0, False, zk, wt, mask, ctx, pkey=pk), "masked")):
        g = VEC[f"proof_{name}_{tag}"]
        assert (pr.pi_a.hex(), pr.pi_b.hex(), pr.pi_c.hex()) == (g["pi_a"], g["pi_b"], g["pi_c"])
        assert pr.publicIO == b"".join(fr(x) for x in REF["publicIO"])
        assert verifyProof(extractVKey(zk), pr, ctx)
    pk.destroy()


def test_pairing_fixture(ctx):
    """e(gen1, gen2): 6 x Fp2 over w^k -> the degree-12 polynomial basis of the fixture ((a + b u) w^k = (a - 9b) w^k + b w^(k+6))"""
    mont = lambda x: (int(x, 16) * (1 << 256) % P).to_bytes(32, "little")   # noqa: E731
    c = REF["constants"]
    g1 = mont(c["gen1"][0]) + mont(c["gen1"][1])
    g2 = b"".join(mont(v) for coord in c["gen2"] for v in coord)       # (x.c0, x.c1, y.c0, y.c1)
    raw = ctx.pairing(g1, g2)
    rinv = pow(1 << 256, -1, P)
    poly = [0] * 12
    for k in range(6):
        a = int.from_bytes(raw[64 * k:64 * k + 32], "little") * rinv % P
        b = int.from_bytes(raw[64 * k + 32:64 * k + 64], "little") * rinv % P
        poly[k] = (poly[k] + a - 9 * b) % P
        poly[k + 6] = (poly[k + 6] + b) % P
    assert [hex(c) for c in poly] == VEC["pairing_gen1_gen2_poly12"]


def test_poseidon_shape_fixture(ctx):
    """tests/golden/poseidon_shape.json: the row-balanced buildABC (rows of 1..13 terms, A and B binned separately, value
    dictionary) and the proof of the small Poseidon-shaped circuit against frozen oracle outputs -- no oracle on the box"""
    import hashlib
    from nim_groth16_amd import Mask, Witness, extractVKey, generateProofWithMask, loadProvingKey, verifyProof
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import poseidonMerkle
    fx = json.load(open(os.path.join(G, "poseidon_shape.json")))
    r1cs, wit = poseidonMerkle(**fx["args"])
    wb = b"".join(fr(x) for x in wit)
    assert wb.hex() == fx["witness_mont"]
    tw = ToxicWaste(*(int(VEC["toxic_waste"][k], 16) for k in ("alpha", "beta", "gamma", "delta", "tau")))
    zk = fakeCircuitSetup(r1cs, tw, 1, ctx)
    assert len(zk.coeffs) == fx["ncoeffs"]
    pk = loadProvingKey(zk, ctx)
    try:
        sha = lambda raw: hashlib.sha256(raw).hexdigest()                  # noqa: E731
        for mont in (True, False):
            w = wb if mont else b"".join((x % R).to_bytes(32, "little") for x in wit)
            Az, Bz, Cz = pk.build_abc(w, mont=mont)
            assert (sha(Az), sha(Bz), sha(Cz)) == (fx["sha256_Az"], fx["sha256_Bz"], fx["sha256_Cz"]), mont
        mask = Mask(int(VEC["mask"]["r"], 16), int(VEC["mask"]["s"], 16))
        pr = generateProofWithMask(0, False, zk, Witness("bn128", len(wit), wb), mask, ctx, pkey=pk)
        g = fx["proof_snarkjs_masked"]
        assert (pr.pi_a.hex(), pr.pi_b.hex(), pr.pi_c.hex()) == (g["pi_a"], g["pi_b"], g["pi_c"])
        assert verifyProof(extractVKey(zk), pr, ctx)
    finally:
        pk.destroy()
