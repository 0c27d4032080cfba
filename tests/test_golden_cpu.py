"""The oracles against the committed fixtures of tests/golden/: the data the reference itself defines
(reference_fixture.json) and the frozen oracle vectors (oracle_vectors.json, made by make_golden.py)."""
import json
import os

from oracle import bn254_ref as o

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = json.load(open(os.path.join(G, "reference_fixture.json")))
VEC = json.load(open(os.path.join(G, "oracle_vectors.json")))


def _frs(hexstr):
    raw = bytes.fromhex(hexstr)
    return [o.fr_from_mont_bytes(raw[i:i + 32]) for i in range(0, len(raw), 32)]


def test_reference_constants_and_toy_anchors():
    c = REF["constants"]
    assert o.P == int(c["primeP"], 16) and o.R == int(c["primeR"], 16)
    assert o.ONE_HALF_FR == int(c["oneHalfFr"], 16) and o.GEN28 == int(c["gen28"], 16)
    assert o.Domain(1 << 20).domainGen == int(c["omega_2p20"], 16)
    assert o.GEN1 == tuple(int(v, 16) for v in c["gen1"])
    assert o.GEN2 == tuple(tuple(int(v, 16) for v in coord) for coord in c["gen2"])
    toy = o.toy_r1cs()
    t = REF["toy_circuit"]
    assert (toy.nWires, toy.nPubOut, toy.nPubIn, toy.nPrivIn) == (t["nWires"], t["nPubOut"], t["nPubIn"], t["nPrivIn"])
    cons = [tuple([(w, int(v) % o.R) for w, v in part] for part in con) for con in t["constraints"]]
    assert [tuple([(w, v % o.R) for w, v in part] for part in con) for con in toy.constraints] == cons
    assert o.TOY_WITNESS == t["witness"]
    Az, Bz, Cz = o.build_abc(o.r1cs_to_coeffs(toy), REF["domain_size"], o.TOY_WITNESS)
    assert (Az, Bz, Cz) == (REF["Az"], REF["Bz"], REF["Cz"])
    qs = o.compute_snarkjs_scalar_coeffs(Az, Bz, Cz)
    for k, v in REF["snarkjs_qs"].items():
        assert qs[int(k)] == int(v, 16)


def test_oracle_still_produces_the_frozen_vectors():
    for group, C, dec in ((1, o.G1, o.g1_from_bytes), (2, o.G2, o.g2_from_bytes)):
        v = VEC[f"msm_g{group}"]
        psz = 64 * group
        pts = bytes.fromhex(v["points"])
        acc = C.inf
        for s, i in zip(_frs(v["scalars"]), range(v["n"])):
            acc = C.add(acc, C.mul(s, dec(pts[psz * i:psz * (i + 1)])))
        assert acc == dec(bytes.fromhex(v["result"]))
    xs = _frs(VEC["ntt16"]["input"])
    assert o.forward_ntt(xs, o.Domain(16)) == _frs(VEC["ntt16"]["forward"])
    assert o.inverse_ntt(xs, o.Domain(16)) == _frs(VEC["ntt16"]["inverse"])
    assert o.compute_snarkjs_scalar_coeffs(REF["Az"], REF["Bz"], REF["Cz"]) == _frs(VEC["quotient_toy"]["snarkjs"])
    assert o.compute_quotient_pointwise(REF["Az"], REF["Bz"], REF["Cz"]) == _frs(VEC["quotient_toy"]["jensgroth"])
    tw = [int(VEC["toxic_waste"][k], 16) for k in ("alpha", "beta", "gamma", "delta", "tau")]
    r, s = int(VEC["mask"]["r"], 16), int(VEC["mask"]["s"], 16)
    for flavour, name in ((o.SNARKJS, "snarkjs"), (o.JENS_GROTH, "jensgroth")):
        zk = o.fake_circuit_setup(o.toy_r1cs(), o.ToxicWaste(*tw), flavour)
        for (rr, ss), tag in (((0, 0), "trivial_mask"), ((r, s), "masked")):
            pr = o.generate_proof_with_mask(zk, o.TOY_WITNESS, rr, ss)
            g = VEC[f"proof_{name}_{tag}"]
            assert (o.g1_to_bytes(pr.pi_a).hex(), o.g2_to_bytes(pr.pi_b).hex(), o.g1_to_bytes(pr.pi_c).hex()) == \
                (g["pi_a"], g["pi_b"], g["pi_c"])
    assert [hex(c) for c in o.pairing(o.GEN1, o.GEN2)] == VEC["pairing_gen1_gen2_poly12"]


def test_c_oracle_against_the_frozen_vectors(orc):
    for group in (1, 2):
        v = VEC[f"msm_g{group}"]
        assert orc.msm(group, bytes.fromhex(v["scalars"]), bytes.fromhex(v["points"])).hex() == v["result"]
        assert orc.msm_naive(group, bytes.fromhex(v["scalars"]), bytes.fromhex(v["points"])).hex() == v["result"]
    n = VEC["ntt16"]
    assert orc.ntt(bytes.fromhex(n["input"]), 4, False).hex() == n["forward"]
    assert orc.ntt(bytes.fromhex(n["input"]), 4, True).hex() == n["inverse"]
    abc = [b"".join(o.fr_to_mont_bytes(x % o.R) for x in REF[k]) for k in ("Az", "Bz", "Cz")]
    assert orc.quotient_snarkjs(abc[0], abc[1], abc[2], 3).hex() == VEC["quotient_toy"]["snarkjs"]


def test_poseidon_shape_generator_and_oracle_still_produce_the_fixture():
    """tests/golden/poseidon_shape.json: the Poseidon-shaped circuit generator (the benchmark's second workload at 2^20)
    still emits the same circuit and witness, and the oracle the same Az / Bz / Cz for them.  (The frozen proof is
    re-derived by make_golden.py and held against the GPU by tests/test_gpu_golden.py.)"""
    import hashlib
    from nim_groth16_amd.synthetic import checkWitness, poseidonMerkle
    fx = json.load(open(os.path.join(G, "poseidon_shape.json")))
    r1cs, wit = poseidonMerkle(**fx["args"])
    assert (r1cs.nWires, r1cs.nConstraints) == (fx["nWires"], fx["nConstraints"]) and checkWitness(r1cs, wit)
    enc = lambda xs: b"".join(o.fr_to_mont_bytes(x) for x in xs)          # noqa: E731
    sha = lambda raw: hashlib.sha256(raw).hexdigest()                      # noqa: E731
    assert enc(wit).hex() == fx["witness_mont"] and sha(enc(wit)) == fx["sha256_witness_mont"] and hex(wit[1]) == fx["root"]
    oc = o.r1cs_to_coeffs(o.R1CS(r1cs.nWires, r1cs.nPubOut, r1cs.nPubIn, r1cs.nPrivIn, r1cs.constraints))
    assert len(oc) == fx["ncoeffs"]
    raw = b"".join(sorted(bytes([m]) + row.to_bytes(4, "little") + col.to_bytes(4, "little") + o.fr_to_mont_bytes(v)
                          for (m, row, col, v) in oc))
    assert sha(raw) == fx["sha256_sorted_coeffs"]
    cons = r1cs.constraints
    assert max(len(c[0]) for c in cons) == fx["max_terms_A"] and max(len(c[1]) for c in cons) == fx["max_terms_B"]
    Az, Bz, Cz = o.build_abc(oc, 1 << fx["args"]["log2n"], wit)
    assert (sha(enc(Az)), sha(enc(Bz)), sha(enc(Cz))) == (fx["sha256_Az"], fx["sha256_Bz"], fx["sha256_Cz"])
