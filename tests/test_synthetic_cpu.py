"""The Poseidon-shaped Merkle-inclusion circuit (nim_groth16_amd/synthetic.py: BASELINE config 5's workload shape built
without circom) on the CPU: the generated witness satisfies every constraint, the rows have the length spread the
row-balanced buildABC kernel is designed for, and the numpy record form of ZKey.coeffs equals the list form."""
import numpy as np

from oracle import bn254_ref as o


def test_poseidon_merkle_witness_satisfies_every_constraint():
    from nim_groth16_amd.synthetic import checkWitness, poseidonMerkle
    for log2n, kw in ((8, dict(full=2, partial=5)), (10, {}), (11, dict(cap=9))):
        r1cs, wit = poseidonMerkle(log2n, seed=4, **kw)
        assert o.ceiling_log2(r1cs.nConstraints + 2) == log2n            # fake_setup.nim:203-206
        assert len(wit) == r1cs.nWires and wit[0] == 1
        assert checkWitness(r1cs, wit)
        bad = list(wit)
        bad[r1cs.nWires // 2] = (bad[r1cs.nWires // 2] + 1) % o.R
        assert not checkWitness(r1cs, bad)


def test_poseidon_merkle_has_the_row_shape_of_a_circom_circuit():
    """rows of 1 .. ~25 terms, ncoeffs >> n, and a third of the wires absent from B (their pointsB1 / pointsB2 are the
    point at infinity in a key, zkey.nim loads them as (0,0): curves.nim:95-98)"""
    from nim_groth16_amd.synthetic import poseidonMerkle
    r1cs, _ = poseidonMerkle(12, seed=4)
    la, lb = r1cs.rowLengths("A"), r1cs.rowLengths("B")
    assert la.min() == 1 and la.max() >= 24 and lb.max() >= 24
    assert (la + lb).sum() / (1 << 12) > 10                              # squaringChain: 2
    assert ((la >= 3) & (la <= 30)).sum() > 0.3 * r1cs.nConstraints
    in_b = np.zeros(r1cs.nWires, dtype=bool)
    in_b[r1cs.B[1]] = True
    assert 0.25 < 1 - in_b.mean() < 0.40
    in_a = np.zeros(r1cs.nWires, dtype=bool)
    in_a[r1cs.A[1]] = True
    assert in_a[2:].all()


def test_coeff_array_equals_the_list_form():
    """r1csToCoeffArray (vectorised) holds the same multiset of (matrix, row, col, value) as r1csToCoeffs on the list
    form of the same circuit (fake_setup.nim:46-65), and packs to the same g16_coeff records"""
    from nim_groth16_amd.fake_setup import R1CS, r1csToCoeffArray, r1csToCoeffs
    from nim_groth16_amd.synthetic import poseidonMerkle
    from nim_groth16_amd.zkey_types import packCoeffs
    r1cs, _ = poseidonMerkle(9, seed=7, full=2, partial=9)
    arr = r1csToCoeffArray(r1cs)
    lst = r1csToCoeffs(R1CS(r1cs.nWires, r1cs.nPubOut, r1cs.nPubIn, r1cs.nPrivIn, r1cs.constraints))
    assert len(arr) == len(lst) and sorted(arr) == sorted(lst) and arr[3] == list(arr)[3]
    rec = lambda b: sorted(b[i:i + 48] for i in range(0, len(b), 48))        # noqa: E731
    assert rec(packCoeffs(arr)) == rec(packCoeffs(lst))
    # the oracle's own restatement of r1csToCoeffs agrees
    oc = o.r1cs_to_coeffs(o.R1CS(r1cs.nWires, 1, 0, r1cs.nPrivIn, r1cs.constraints))
    assert sorted((m, r, c, o.fr_from_mont_bytes(v)) for (m, r, c, v) in arr) == sorted(oc)


def test_c_oracle_build_abc_on_long_rows(orc):
    """orc_build_abc (the checker of the GPU buildABC at 2^18) against the Python transliteration of prover.nim:56-73
    on rows of up to 25 terms"""
    from nim_groth16_amd.fake_setup import r1csToCoeffArray
    from nim_groth16_amd.synthetic import poseidonMerkle
    from nim_groth16_amd.zkey_types import packCoeffs
    from tests import inputs as I
    r1cs, wit = poseidonMerkle(10, seed=4)
    arr = r1csToCoeffArray(r1cs)
    Az, Bz, Cz = orc.build_abc(packCoeffs(arr), I.fr_mont_bytes(wit), 10)
    eA, eB, eC = o.build_abc([(m, r, c, o.fr_from_mont_bytes(v)) for (m, r, c, v) in arr], 1 << 10, wit)
    assert I.fr_from_mont(Az) == eA and I.fr_from_mont(Bz) == eB and I.fr_from_mont(Cz) == eC
    assert eC[:r1cs.nConstraints] != [0] * r1cs.nConstraints
