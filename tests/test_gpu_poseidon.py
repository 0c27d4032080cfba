"""BASELINE config 5's workload shape (a Poseidon-shaped Merkle-inclusion circuit: rows of 1..25 terms, ncoeffs ~ 12 n,
a third of the wires absent from B) through the C ABI: the row-balanced sparse kernel (g16_spmv_fr, g16_build_abc) and
full proofs, bit for bit against the oracle.  buildABC: reference groth16/prover.nim:56-73 over the entries of
files/zkey.nim:169-192; column dot products of the setup: fake_setup.nim:159-187."""
import numpy as np
import pytest

from oracle import bn254_ref as o
from tests import inputs as I

pytestmark = pytest.mark.gpu
R = o.R


def _toxic(seed=5):
    from nim_groth16_amd.fake_setup import ToxicWaste
    from nim_groth16_amd.synthetic import SplitMix64
    rng = SplitMix64(seed)
    return ToxicWaste(*[rng.fr() for _ in range(5)]), rng


def test_spmv_against_python_ints(ctx):
    """g16_spmv_fr on a matrix with every row class of the kernel: empty rows, rows of 1..4 terms (one lane), 5..256
    (groups of 2..64 lanes), a 5000-term row (a looping wave), duplicate (row, col) pairs, and both value paths -- a
    handful of distinct values (dictionary) and all-distinct values (plain)"""
    from nim_groth16_amd import bn128 as F
    rng = o.SplitMix64(91)
    ncols, nrows = 777, 300
    x = [rng.fr() for _ in range(ncols)]
    lens = [0, 1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32, 33, 64, 65, 128, 129, 255, 256, 257, 1000, 5000]
    for few in (True, False):
        pool = [rng.fr() for _ in range(5)] + [1, R - 1]
        row, col, val = [], [], []
        for r in range(nrows):
            L = lens[r] if r < len(lens) else rng.next() % 40
            for _ in range(L):
                row.append(r), col.append(rng.next() % ncols)
                val.append(pool[rng.next() % len(pool)] if few else rng.fr())
        row += [5, 5]                                                     # duplicates add up
        col += [3, 3]
        val += [7, 9]
        perm = np.random.default_rng(1).permutation(len(row))             # triplets in any order
        row, col, val = (np.array(a, dtype=object)[perm] for a in (row, col, val))
        want = [0] * nrows
        for r, c, v in zip(row, col, val):
            want[r] = (want[r] + v * x[c]) % R
        vb = np.frombuffer(F.frSeqToMontBytes(val), dtype=np.uint8).reshape(-1, 32)
        got = ctx.spmv(row.astype(np.uint32), col.astype(np.uint32), vb, F.frSeqToMontBytes(x), nrows)
        assert F.frSeqFromMontBytes(got) == want
    assert ctx.spmv(np.zeros(0, np.uint32), np.zeros(0, np.uint32), b"", F.frSeqToMontBytes(x), 3) == bytes(96)
    from nim_groth16_amd._lib import G16Error
    with pytest.raises(G16Error):                                         # column out of range
        ctx.spmv(np.array([0], np.uint32), np.array([ncols], np.uint32), bytes(32), F.frSeqToMontBytes(x), 3)


def test_poseidon_2p10_key_equals_the_oracle_setup(ctx, orc):
    """fakeCircuitSetup with the column dot products on the GPU (fake_setup.nim:159-187, 254-256 through
    g16_spmv_fr) == the oracle's transliteration of fake_setup.nim:201-326 on the list form, every array"""
    from nim_groth16_amd.fake_setup import fakeCircuitSetup
    from nim_groth16_amd.synthetic import poseidonMerkle
    from tests.test_gpu_prover import _chunks, _zkey_to_oracle
    r1cs, wit = poseidonMerkle(10, seed=4)
    tox, _ = _toxic()
    for flavour in (1, 0):
        zk = fakeCircuitSetup(r1cs, tox, flavour, ctx)
        bg1 = lambda ks: [o.g1_from_bytes(x) for x in _chunks(orc.fixed_base(1, I.fr_mont_bytes(ks)), 64)]    # noqa: E731
        bg2 = lambda ks: [o.g2_from_bytes(x) for x in _chunks(orc.fixed_base(2, I.fr_mont_bytes(ks)), 128)]   # noqa: E731
        oz = o.fake_circuit_setup(o.R1CS(r1cs.nWires, 1, 0, r1cs.nPrivIn, r1cs.constraints),
                                  o.ToxicWaste(tox.alpha, tox.beta, tox.gamma, tox.delta, tox.tau),
                                  o.SNARKJS if flavour else o.JENS_GROTH, bg1, bg2)
        z2 = _zkey_to_oracle(zk)
        for f in ("alpha1", "beta1", "delta1", "beta2", "gamma2", "delta2", "pointsIC", "pointsA1", "pointsB1",
                  "pointsB2", "pointsC1", "pointsH1"):
            assert getattr(z2, f) == getattr(oz, f), f
        assert sorted(z2.coeffs) == sorted(oz.coeffs)


def _poseidon_key(ctx, log2n):
    from nim_groth16_amd import loadProvingKey
    from nim_groth16_amd.fake_setup import fakeCircuitSetup
    from nim_groth16_amd.synthetic import poseidonMerkle
    r1cs, wit = poseidonMerkle(log2n, seed=4)
    tox, rng = _toxic()
    zk = fakeCircuitSetup(r1cs, tox, 1, ctx)
    assert zk.header.domainSize == 1 << log2n
    return r1cs, wit, zk, loadProvingKey(zk, ctx), rng


def test_build_abc_2p18_every_row_vs_c_oracle_and_full_proof_bit_exact(ctx, orc):
    """g16_build_abc == the C oracle's buildABC on every row of a 2^18 Poseidon-shaped circuit (3.2 M entries, rows of
    1..25 terms), for the Nim seq[Fr] witness and the raw .wtns layout; then generateProofWithMask (prover.nim:215-304)
    bit-exact vs the C oracle's proof + the pairing equation + the GPU verifier"""
    from nim_groth16_amd import Mask, Witness, generateProofWithMask
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.zkey_types import packCoeffs
    from tests.parity import check_gpu_proof
    log2n = 18
    r1cs, wit, zk, pk, rng = _poseidon_key(ctx, log2n)
    try:
        info = pk.abc_info()
        assert info["ncoeffs"] == len(zk.coeffs) > 10 << log2n
        assert 0 < info["dict_values"] < 1000                       # MDS entries, round constants, +-1
        g = info["rows_by_terms"]           # rows of A and of B, binned separately
        assert sum(g.values()) == 2 << log2n and min(g[k] for k in ("L<=1", "L=2", "L<=4", "L<=8", "L<=16", "L<=32")) > 0
        assert g["L<=64"] == g["L<=128"] == g["L>128"] == 0
        wb = F.frSeqToMontBytes(wit)
        want = orc.build_abc(packCoeffs(zk.coeffs), wb, log2n)
        assert pk.build_abc(wb) == want
        assert pk.build_abc(F.frSeqToStdBytes(wit), mont=False) == want
        inf = pk.inf_counts()
        assert 0.25 < inf["B1"] / zk.header.nvars < 0.40 and inf["B1"] == inf["B2"] == inf["B1_and_B2"]
        assert inf["compact_B"] and not inf["compact_A"] and inf["A1"] <= 2
        mask = Mask(rng.fr(), rng.fr())
        pr = generateProofWithMask(0, False, zk, Witness("bn128", len(wit), wb), mask, ctx, pkey=pk)
        check_gpu_proof(orc, zk, wit, wb, mask.r, mask.s, (pr.pi_a, pr.pi_b, pr.pi_c), ctx)
        std = generateProofWithMask(0, False, zk, Witness("bn128", len(wit), F.frSeqToStdBytes(wit), std=True), mask,
                                    ctx, pkey=pk)
        assert (std.pi_a, std.pi_b, std.pi_c) == (pr.pi_a, pr.pi_b, pr.pi_c)
    finally:
        pk.destroy()


def test_build_abc_rows_longer_than_a_wave(ctx, orc):
    """a key whose A matrix has rows of 300 and 3000 terms and whose B has a 70-term row (the looping 64-lane group),
    all-distinct values (no dictionary): g16_build_abc == C oracle"""
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd import loadProvingKey
    from nim_groth16_amd.fake_setup import R1CS, fakeCircuitSetup
    from nim_groth16_amd.zkey_types import packCoeffs
    rng = o.SplitMix64(17)
    nw, m = 4000, 62
    cons = []
    for i in range(m):
        la = 3000 if i == 7 else 300 if i == 20 else 1 + rng.next() % 6
        lb = 70 if i == 9 else 1 + rng.next() % 3
        cons.append(([(2 + rng.next() % (nw - 2), rng.fr()) for _ in range(la)],
                     [(2 + rng.next() % (nw - 2), rng.fr()) for _ in range(lb)], []))
    wit = [1] + [rng.fr() for _ in range(nw - 1)]
    tox, _ = _toxic()
    zk = fakeCircuitSetup(R1CS(nw, 1, 0, nw - 2, cons), tox, 1, ctx)
    pk = loadProvingKey(zk, ctx)
    try:
        info = pk.abc_info()
        assert info["dict_values"] == 0 and info["rows_by_terms"]["L>128"] == 2 and info["rows_by_terms"]["L<=128"] == 1
        wb = F.frSeqToMontBytes(wit)
        assert pk.build_abc(wb) == orc.build_abc(packCoeffs(zk.coeffs), wb, 6)
    finally:
        pk.destroy()
