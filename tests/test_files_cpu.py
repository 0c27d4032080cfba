"""Host-side file formats (mirrors of groth16/files/*.nim): writer -> parser round trips on the reference's toy
circuit, byte-level checks of the layout facts the GPU path relies on, JSON export shape.  No GPU needed: the
key comes from the oracle's fake setup."""
import json
import struct

import pytest

from oracle import bn254_ref as o
from tests import inputs as I


def _toy_zkey():
    from nim_groth16_amd.zkey_types import GrothHeader, ProverPoints, SpecPoints, ZKey
    rng = o.SplitMix64(123)
    oz = o.fake_circuit_setup(o.toy_r1cs(), o.ToxicWaste(*[rng.fr() for _ in range(5)]), o.SNARKJS)
    g1 = lambda ps: b"".join(o.g1_to_bytes(p) for p in ps)      # noqa: E731
    g2 = lambda ps: b"".join(o.g2_to_bytes(p) for p in ps)      # noqa: E731
    zk = ZKey(GrothHeader("bn128", 1, oz.nvars, oz.npubs, oz.domainSize, oz.logDomainSize),
              SpecPoints(o.g1_to_bytes(oz.alpha1), o.g1_to_bytes(oz.beta1), o.g2_to_bytes(oz.beta2),
                         o.g2_to_bytes(oz.gamma2), o.g1_to_bytes(oz.delta1), o.g2_to_bytes(oz.delta2)),
              g1(oz.pointsIC), ProverPoints(g1(oz.pointsA1), g1(oz.pointsB1), g2(oz.pointsB2), g1(oz.pointsC1),
                                            g1(oz.pointsH1)),
              [(m, r, c, o.fr_to_mont_bytes(v)) for (m, r, c, v) in oz.coeffs])
    return zk, oz


def test_zkey_roundtrip_and_layout(tmp_path):
    from nim_groth16_amd.files import parseZKey, writeZKey
    zk, oz = _toy_zkey()
    path = str(tmp_path / "toy.zkey")
    writeZKey(path, zk)
    back = parseZKey(path)
    assert back == zk
    raw = open(path, "rb").read()
    assert raw[:4] == b"zkey" and struct.unpack_from("<II", raw, 4) == (1, 9)      # container.nim:6-20
    # section 4 stores c * R^2 (double Montgomery, zkey.nim:57): the last dummy coefficient is 1 -> R^2 mod r
    assert (o.MONT * o.MONT % o.R).to_bytes(32, "little") in raw
    # point sections are the in-memory Montgomery layout: gen-derived point bytes appear verbatim
    assert zk.pPoints.pointsA1 in raw and zk.pPoints.pointsB2 in raw


def test_zkey_raw_coefficient_section(tmp_path):
    """parseZKey(rawCoeffs=True) keeps section 4 as it lies in the file (for g16_pkey_create_zkey): the same entries as
    the parsed form, values still in the double-Montgomery form c R^2 (io.nim:134-139)"""
    from nim_groth16_amd.files import parseZKey, writeZKey
    zk, _ = _toy_zkey()
    path = str(tmp_path / "toy.zkey")
    writeZKey(path, zk)
    raw = parseZKey(path, rawCoeffs=True)
    s4 = raw.coeffsSection4
    assert raw.coeffs == [] and struct.unpack_from("<I", s4, 0)[0] == len(zk.coeffs) and len(s4) == 4 + 44 * len(zk.coeffs)
    for i, (m, r, c, v) in enumerate(zk.coeffs):
        assert struct.unpack_from("<III", s4, 4 + 44 * i) == (m, r, c)
        file_int = int.from_bytes(s4[4 + 44 * i + 12: 4 + 44 * i + 44], "little")
        assert file_int == int.from_bytes(v, "little") * o.MONT % o.R            # c R  ->  c R^2
    assert raw.pPoints == zk.pPoints and raw.header == zk.header and raw.specPoints == zk.specPoints


def test_zkey_parser_rejects_bad_files(tmp_path):
    from nim_groth16_amd.files import parseZKey, writeZKey
    zk, _ = _toy_zkey()
    path = str(tmp_path / "bad.zkey")
    writeZKey(path, zk)
    raw = bytearray(open(path, "rb").read())
    raw[0:4] = b"wtns"
    open(path, "wb").write(raw)
    with pytest.raises(AssertionError):
        parseZKey(path)                                                  # container.nim:85


def test_witness_roundtrip(tmp_path):
    from nim_groth16_amd.files import parseWitness, writeWitness
    path = str(tmp_path / "toy.wtns")
    writeWitness(path, o.TOY_WITNESS)
    w = parseWitness(path)
    assert w.curve == "bn128" and w.nvars == 8 and w.std
    assert w.values == I.fr_std_bytes(o.TOY_WITNESS)                     # standard form, witness.nim:14
    assert open(path, "rb").read()[:4] == b"wtns"


def test_r1cs_roundtrip(tmp_path):
    from nim_groth16_amd.fake_setup import R1CS
    from nim_groth16_amd.files import parseR1CS, writeR1CS
    toy = o.toy_r1cs()
    r1 = R1CS(8, 1, 1, 3, toy.constraints)
    path = str(tmp_path / "toy.r1cs")
    writeR1CS(path, r1)
    back = parseR1CS(path)
    assert (back.nWires, back.nPubOut, back.nPubIn, back.nPrivIn) == (8, 1, 1, 3)
    assert back.constraints == [tuple([(i, v % o.R) for (i, v) in lc] for lc in c) for c in toy.constraints]
    assert back.wireToLabel == list(range(8))


def test_export_json(tmp_path):
    from nim_groth16_amd.files import exportProof, exportPublicIO
    from nim_groth16_amd.prover import Proof
    _, oz = _toy_zkey()
    ref = o.generate_proof_with_mask(oz, o.TOY_WITNESS, 11, 22)
    prf = Proof(I.fr_mont_bytes(ref.publicIO), o.g1_to_bytes(ref.pi_a), o.g2_to_bytes(ref.pi_b), o.g1_to_bytes(ref.pi_c))
    pj, ij = str(tmp_path / "proof.json"), str(tmp_path / "public.json")
    exportProof(pj, prf)
    exportPublicIO(ij, prf)
    d = json.load(open(pj))
    assert d["protocol"] == "groth16" and d["curve"] == "bn128"
    assert [int(x) for x in d["pi_a"]] == [ref.pi_a[0], ref.pi_a[1], 1]                     # export_json.nim:55-59
    assert [[int(x) for x in row] for row in d["pi_b"]] == [list(ref.pi_b[0]), list(ref.pi_b[1]), [1, 0]]
    assert [int(x) for x in d["pi_c"]] == [ref.pi_c[0], ref.pi_c[1], 1]
    assert [int(x) for x in json.load(open(ij))] == [2023, 1022]                            # constant 1 skipped


def test_export_vkey_json(tmp_path):
    """snarkjs verification_key.json shape (an addition: the reference has no JSON key export)"""
    from nim_groth16_amd.files import exportVKey
    from nim_groth16_amd.verifier import extractVKey
    zk, oz = _toy_zkey()
    path = str(tmp_path / "vkey.json")
    exportVKey(path, extractVKey(zk))
    d = json.load(open(path))
    assert d["protocol"] == "groth16" and d["curve"] == "bn128" and d["nPublic"] == 2 and len(d["IC"]) == 3
    assert [int(x) for x in d["vk_alpha_1"]] == [oz.alpha1[0], oz.alpha1[1], 1]
    assert [[int(x) for x in row] for row in d["vk_delta_2"]] == [list(oz.delta2[0]), list(oz.delta2[1]), [1, 0]]
    assert [[int(x) for x in row] for row in d["vk_gamma_2"]] == [list(oz.gamma2[0]), list(oz.gamma2[1]), [1, 0]]
    assert [int(x) for x in d["IC"][2]] == [oz.pointsIC[2][0], oz.pointsIC[2][1], 1]


def _snarkjs_like(tmp_path, order=(1, 2, 4, 3, 9, 8, 5, 6, 7, 10), shuffle_coeffs=False):
    """oracle fake setup of the unused-wire circuit, serialised by tests/snarkjs_layout.py (NOT by writeZKey)"""
    from tests import snarkjs_layout as L
    nw, npo, npi, npriv, cons, wit = L.unused_wire_circuit()
    rng = o.SplitMix64(321)
    oz = o.fake_circuit_setup(o.R1CS(nw, npo, npi, npriv, cons), o.ToxicWaste(*[rng.fr() for _ in range(5)]), o.SNARKJS)
    g1 = lambda ps: b"".join(o.g1_to_bytes(p) for p in ps)      # noqa: E731
    g2 = lambda ps: b"".join(o.g2_to_bytes(p) for p in ps)      # noqa: E731
    coeffs = list(oz.coeffs)
    if shuffle_coeffs:
        import random
        random.Random(7).shuffle(coeffs)
    spec = (o.g1_to_bytes(oz.alpha1), o.g1_to_bytes(oz.beta1), o.g2_to_bytes(oz.beta2), o.g2_to_bytes(oz.gamma2),
            o.g1_to_bytes(oz.delta1), o.g2_to_bytes(oz.delta2))
    raw = L.snarkjs_zkey_bytes(oz.nvars, oz.npubs, oz.domainSize, spec, g1(oz.pointsIC), g1(oz.pointsA1), g1(oz.pointsB1),
                               g2(oz.pointsB2), g1(oz.pointsC1), g1(oz.pointsH1), coeffs, order)
    zpath, wpath = str(tmp_path / "snarkjs_like.zkey"), str(tmp_path / "snarkjs_like.wtns")
    open(zpath, "wb").write(raw)
    open(wpath, "wb").write(L.snarkjs_wtns_bytes(wit))
    return zpath, wpath, oz, wit, coeffs


@pytest.mark.parametrize("order,shuffle", [((1, 2, 4, 3, 9, 8, 5, 6, 7, 10), False), ((10, 9, 8, 7, 6, 5, 4, 3, 2, 1), True),
                                           ((1, 2, 3, 4, 5, 6, 7, 8, 9), False)])
def test_parse_zkey_laid_out_like_snarkjs(tmp_path, order, shuffle):
    """sections out of order, a section 10, infinity points on unused wires, the dummy public rows: parsed into
    exactly the key the oracle's setup produced (format authority: groth16/files/zkey.nim:6-91, container.nim:6-20)"""
    from nim_groth16_amd.files import parseWitness, parseZKey
    zpath, wpath, oz, wit, coeffs = _snarkjs_like(tmp_path, order, shuffle)
    zk = parseZKey(zpath)
    h = zk.header
    assert (h.curve, h.flavour, h.nvars, h.npubs, h.domainSize, h.logDomainSize) == ("bn128", 1, 5, 1, 4, 2)
    dec1 = lambda b: [o.g1_from_bytes(b[i:i + 64]) for i in range(0, len(b), 64)]       # noqa: E731
    dec2 = lambda b: [o.g2_from_bytes(b[i:i + 128]) for i in range(0, len(b), 128)]     # noqa: E731
    pp = zk.pPoints
    assert dec1(pp.pointsA1) == oz.pointsA1 and dec1(pp.pointsB1) == oz.pointsB1 and dec2(pp.pointsB2) == oz.pointsB2
    assert dec1(pp.pointsC1) == oz.pointsC1 and dec1(pp.pointsH1) == oz.pointsH1 and dec1(zk.pointsIC) == oz.pointsIC
    assert dec1(zk.specPoints.alpha1)[0] == oz.alpha1 and dec2(zk.specPoints.gamma2)[0] == oz.gamma2
    # infinity = (0,0) for wires that do not occur in A / in B (curves.nim:49-50; snarkjs writes zero bytes)
    assert pp.pointsA1[64 * 3:64 * 4] == bytes(64) and dec1(pp.pointsA1)[3] == o.INF_G1      # wire b: only in B
    assert pp.pointsB2[128 * 2:128 * 3] == bytes(128) and pp.pointsB1[64 * 2:64 * 3] == bytes(64)   # wire a: only in A
    # wire c (public) occurs in no constraint's A or B -- but the dummy row `1 * w_1` gives it an A point, no B point
    assert pp.pointsA1[64:128] != bytes(64) and pp.pointsB2[128:256] == bytes(128) and pp.pointsB1[64:128] == bytes(64)
    # section 4: double-Montgomery values decoded to single Montgomery; the npubs+1 dummy rows `1 * w_i` of matrix A
    # sit behind the 2 real constraints (fake_setup.nim:59-63)
    assert [(m, r, c, o.fr_from_mont_bytes(v)) for (m, r, c, v) in zk.coeffs] == coeffs
    assert {(m, r, c, o.fr_from_mont_bytes(v)) for (m, r, c, v) in zk.coeffs} >= {(0, 2, 0, 1), (0, 3, 1, 1)}
    w = parseWitness(wpath)
    assert w.std and w.nvars == 5 and I.fr_from_mont(I.fr_mont_bytes(wit)) == wit
    assert [int.from_bytes(w.values[32 * i:32 * i + 32], "little") for i in range(5)] == wit


def test_parse_witness_with_sections_swapped(tmp_path):
    from nim_groth16_amd.files import parseWitness
    from tests import snarkjs_layout as L
    path = str(tmp_path / "swapped.wtns")
    open(path, "wb").write(L.snarkjs_wtns_bytes(o.TOY_WITNESS, order=(2, 1)))
    w = parseWitness(path)
    assert w.nvars == 8 and w.values == I.fr_std_bytes(o.TOY_WITNESS)


def test_zkey_parser_rejects_truncated_and_mislabelled_sections(tmp_path):
    from nim_groth16_amd.files import parseZKey
    zpath, _, _, _, _ = _snarkjs_like(tmp_path)
    raw = open(zpath, "rb").read()
    bad = str(tmp_path / "trunc.zkey")
    open(bad, "wb").write(raw[:-100])                     # the last section runs past the end of the file
    with pytest.raises(AssertionError, match="truncated section"):
        parseZKey(bad)
    # a points section one point short: "unexpected section length" (zkey.nim:199 and the like)
    from tests import snarkjs_layout as L
    nw, npo, npi, npriv, cons, wit = L.unused_wire_circuit()
    rng = o.SplitMix64(321)
    oz = o.fake_circuit_setup(o.R1CS(nw, npo, npi, npriv, cons), o.ToxicWaste(*[rng.fr() for _ in range(5)]), o.SNARKJS)
    g1 = lambda ps: b"".join(o.g1_to_bytes(p) for p in ps)      # noqa: E731
    g2 = lambda ps: b"".join(o.g2_to_bytes(p) for p in ps)      # noqa: E731
    spec = (o.g1_to_bytes(oz.alpha1), o.g1_to_bytes(oz.beta1), o.g2_to_bytes(oz.beta2), o.g2_to_bytes(oz.gamma2),
            o.g1_to_bytes(oz.delta1), o.g2_to_bytes(oz.delta2))
    short = L.snarkjs_zkey_bytes(oz.nvars, oz.npubs, oz.domainSize, spec, g1(oz.pointsIC), g1(oz.pointsA1[:-1]),
                                 g1(oz.pointsB1), g2(oz.pointsB2), g1(oz.pointsC1), g1(oz.pointsH1), list(oz.coeffs))
    open(bad, "wb").write(short)
    with pytest.raises(AssertionError, match="unexpected section length"):
        parseZKey(bad)
