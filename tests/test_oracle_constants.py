"""Pins the oracle to every constant the reference embeds (SURVEY.md 8c 'known-answer anchors')."""
from oracle import bn254_ref as o


def test_field_primes_hex_equals_decimal():
    # groth16/bn128/fields.nim:5-6 vs :36-37
    assert o.P == 21888242871839275222246405745257275088696311157297823662689037894645226208583
    assert o.R == 21888242871839275222246405745257275088548364400416034343698204186575808495617


def test_montgomery_constants():
    # groth16/bn128/io.nim:87-92
    assert o.FP_MONT_R == 0x0e0a77c19a07df2f666ea36f7879462c0a78eb28f5c70b3dd35d438dc58f0d9d
    assert o.FP_INV_MONT_R == 0x2e67157159e5c639cf63e9cfb74492d9eb2022850278edf8ed84884a014afa37
    assert o.FR_MONT_R == 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb
    assert o.FR_INV_MONT_R == 0x15ebf95182c5551cc8260de4aeb85d5d090ef5a9e111ec87dc5ba0056db1194e


def test_one_half_and_minus_one():
    assert o.ONE_HALF_FR * 2 % o.R == 1                                   # math/ntt.nim:95
    assert (o.P - 1) == 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd46   # fields.nim:49
    assert (o.R - 1) == 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000000   # fields.nim:50


def test_gen28_order_and_origin():
    # math/domain.nim:26 ; 2-adicity of r-1 is 28
    assert pow(o.GEN28, 1 << 28, o.R) == 1 and pow(o.GEN28, 1 << 27, o.R) != 1
    assert o.GEN28 == pow(5, (o.R - 1) >> 28, o.R)
    assert (o.R - 1) % (1 << 28) == 0 and ((o.R - 1) >> 28) % 2 == 1
    assert o.Domain(1 << 20).domainGen == 0x26125da10a0ed06327508aba06d1e303ac616632dbed349f53422da953337857
    assert o.Domain(1 << 21).domainGen == 0x1ded8980ae2bdd1a4222150e8598fc8c58f50577ca5a5ce3b2c87885fcd0b523


def test_curve_constants():
    assert o.G1.is_on_curve(o.GEN1) and o.G2.is_on_curve(o.GEN2)          # curves.nim:112-124
    assert o.fp2_mul(o.TWIST_B, (9, 1)) == (3, 0)                         # B = 3/(9+u), curves.nim:75-77
    assert o.G1.is_inf(o.G1.mul(o.R, o.GEN1)) and o.G2.is_inf(o.G2.mul(o.R, o.GEN2))   # curves.nim:225-229
    assert o.G1.is_on_curve(o.INF_G1) and o.G2.is_on_curve(o.INF_G2)      # curves.nim:55-57


def test_eta_n_is_minus_one():
    # prover.nim:127-128: invZ1 = 1/(eta^n - 1) with eta = w_(2n)
    for lg in (1, 3, 10):
        n = 1 << lg
        assert pow(o.Domain(2 * n).domainGen, n, o.R) == o.R - 1
