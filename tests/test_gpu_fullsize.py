"""Parity at the BASELINE sizes (2^20), where the Python oracle cannot follow, through size-independent properties:
known discrete logs (MSM), round trip / linearity / direct evaluation at a few indices (NTT), and prove -> verify plus
the C oracle's proof (full prover).  Everything on the GPU goes through the C ABI."""
import pytest

from oracle import bn254_ref as o
from tests import inputs as I

pytestmark = pytest.mark.gpu
R = o.R


def _stream(seed, n):
    from nim_groth16_amd.synthetic import _fr_stream
    return _fr_stream(seed, n)


def _bytes(vals):
    from nim_groth16_amd import bn128 as F
    return F.frSeqToMontBytes(vals)


# sizes 2^15 .. 2^19 walk the merged-bucket fold through 2, 4, 8, 16, 32 reduction slices (64 at 2^20)
@pytest.mark.parametrize("group,log2n", [(1, 20), (2, 18), (2, 20), (1, 15), (1, 16), (1, 17), (1, 19), (2, 15), (2, 16)])
def test_msm_fullsize_known_discrete_logs(ctx, orc, group, log2n):
    """P_i = k_i * G, so that sum s_i P_i = (sum s_i k_i mod r) * G  (SURVEY 8d config 2 check).  The point set is
    produced by the product's fixed-base kernel (the oracle would take minutes at 2^20); the EXPECTED value is the
    oracle's own scalar multiplication of the generator, so the check does not loop back into the product -- and a
    wrong point set could not cancel out: a sample of it is held to the oracle's fixed-base multiplier too."""
    n = 1 << log2n
    ks, sc = _stream(1, n), _stream(2, n)
    sc[5] = 0
    sc[7] = sc[6]                                  # a zero scalar and a repeated one
    pts = ctx.fixed_base(group, _bytes(ks))
    psz = 64 * group
    sample = [0, 1, 2, 12345 % n, n // 2, n - 1]
    assert b"".join(pts[psz * i:psz * (i + 1)] for i in sample) == orc.fixed_base(group, _bytes([ks[i] for i in sample]))
    pts = pts[:psz * 9] + bytes(psz) + pts[psz * 10:]            # an infinity point inside the set
    ks[9] = 0
    e = sum(s * k for s, k in zip(sc, ks)) % R
    gen = o.g1_to_bytes(o.GEN1) if group == 1 else o.g2_to_bytes(o.GEN2)
    exp = orc.mul(group, _bytes([e]), gen)                       # oracle, not ctx.fixed_base
    assert exp == orc.fixed_base(group, _bytes([e]))
    sb = _bytes(sc)
    if log2n < 20 or group == 1:
        assert ctx.msm(group, sb, pts, n) == exp                 # one-shot path (no tables)
    h = ctx.register_points(group, pts, n)
    try:
        assert ctx.msm_points(h, sb) == exp
        # linearity in the scalars: MSM(2 s) == 2 MSM(s)
        two = ctx.msm_points(h, _bytes([2 * s % R for s in sc]))
        assert two == orc.mul(group, _bytes([2 * e % R]), gen)
    finally:
        h.release()


def test_ntt_fullsize_roundtrip_linearity_and_direct_evaluation(ctx):
    log2n, n = 20, 1 << 20
    xs, zs = _stream(11, n), _stream(12, n)
    xb, zb = _bytes(xs), _bytes(zs)
    ys = ctx.ntt(xb, log2n, False)
    assert ctx.ntt(ys, log2n, True) == xb                                    # inverse(forward(x)) == x, incl. 1/n
    yv = I.fr_from_mont(ys[:32]) + I.fr_from_mont(ys[32 * 12345:32 * 12346]) + I.fr_from_mont(ys[32 * (n - 1):])
    w = o.Domain(n).domainGen
    for k, got in zip((0, 12345, n - 1), yv):                                 # y_k = sum_i x_i w^(ik)  (ntt.nim:55-77)
        wk, acc = pow(w, k, R), 0
        for x in reversed(xs):                                                # Horner
            acc = (acc * wk + x) % R
        assert got == acc
    sums = ctx.ntt(_bytes([(a + b) % R for a, b in zip(xs, zs)]), log2n, False)
    yz = ctx.ntt(zb, log2n, False)
    for k in (1, 77777, n // 2, n - 2):                                       # linearity
        a, b, c = (I.fr_from_mont(buf[32 * k:32 * k + 32])[0] for buf in (ys, yz, sums))
        assert (a + b) % R == c


@pytest.mark.parametrize("log2n", [21, 22, 24])
def test_ntt_large_domains_roundtrip(ctx, log2n):
    """domains beyond the benchmark size (config 4 uses 2^22; the library accepts up to 2^27): round trip and
    y_0 = sum x_i (the Montgomery encoding is linear, so the sum can be taken on the raw limbs)"""
    import numpy as np
    n = 1 << log2n
    raw = np.random.default_rng(log2n).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= (1 << 59) - 1                                 # < 2^251 < r: valid residues
    xb = raw.astype("<u8").tobytes()
    yb = ctx.ntt(xb, log2n, False)
    assert ctx.ntt(yb, log2n, True) == xb
    s = sum(int(raw[:, j].astype(object).sum()) << (64 * j) for j in range(4))
    assert int.from_bytes(yb[:32], "little") == s % R


def _prove_and_check(ctx, orc, log2n, bit_exact, flavour=1):
    from nim_groth16_amd import (Mask, Witness, extractVKey, generateProofWithMask, loadProvingKey, verifyProof)
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import SplitMix64, squaringChain
    from tests.parity import check_gpu_proof
    m = (1 << log2n) - 2
    r1cs, wit = squaringChain(m, seed=4)
    rng = SplitMix64(5)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), flavour, ctx)
    pk = loadProvingKey(zk, ctx)
    try:
        wb = _bytes(wit)
        mask = Mask(rng.fr(), rng.fr())
        pr = generateProofWithMask(0, False, zk, Witness("bn128", m + 2, wb), mask, ctx, pkey=pk)
        assert verifyProof(extractVKey(zk), pr, ctx)                          # testProver.nim:65-73 at scale
        if bit_exact:   # buildABC, the H scalars and the five MSMs recomputed by the C oracle, then the mask algebra
            check_gpu_proof(orc, zk, wit, wb, mask.r, mask.s, (pr.pi_a, pr.pi_b, pr.pi_c), ctx)
    finally:
        pk.destroy()


def test_full_proof_2p16_bit_exact_vs_c_oracle_and_2p18_verifies(ctx, orc):
    _prove_and_check(ctx, orc, 16, True)
    _prove_and_check(ctx, orc, 18, False)


def test_full_proof_2p14_jensgroth_bit_exact_vs_c_oracle(ctx, orc):
    """the JensGroth flavour (7 NTTs, prover.nim:118-148) beyond the toy size, against the C oracle"""
    _prove_and_check(ctx, orc, 14, True, flavour=0)


def test_full_proof_2p20_bit_exact_vs_c_oracle(ctx, orc):
    """BASELINE config 3 (2^20-constraint circuit, full prove) inside the suite: bit-exact against the C oracle's
    proof and accepted by both verifiers -- the same check that gates every bench.py run (tests/parity.py)."""
    _prove_and_check(ctx, orc, 20, True)
