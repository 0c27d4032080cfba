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


@pytest.mark.parametrize("group,log2n", [(1, 20), (2, 18)])
def test_msm_fullsize_known_discrete_logs(ctx, group, log2n):
    """P_i = k_i * G (fixed-base kernel), so that sum s_i P_i = (sum s_i k_i mod r) * G  (SURVEY 8d config 2 check)"""
    n = 1 << log2n
    ks, sc = _stream(1, n), _stream(2, n)
    sc[5] = 0
    sc[7] = sc[6]                                  # a zero scalar and a repeated one
    pts = ctx.fixed_base(group, _bytes(ks))
    psz = 64 * group
    pts = pts[:psz * 9] + bytes(psz) + pts[psz * 10:]            # an infinity point inside the set
    ks[9] = 0
    e = sum(s * k for s, k in zip(sc, ks)) % R
    exp = ctx.fixed_base(group, _bytes([e]))
    sb = _bytes(sc)
    assert ctx.msm(group, sb, pts, n) == exp
    h = ctx.register_points(group, pts, n)
    try:
        assert ctx.msm_points(h, sb) == exp
        # linearity in the scalars: MSM(2 s) == 2 MSM(s)
        two = ctx.msm_points(h, _bytes([2 * s % R for s in sc]))
        assert two == ctx.fixed_base(group, _bytes([2 * e % R]))
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


def test_full_proof_2p16_bit_exact_vs_c_oracle_and_2p18_verifies(ctx, orc):
    from nim_groth16_amd import (Mask, Witness, extractVKey, generateProofWithMask, loadProvingKey, verifyProof)
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import SplitMix64, squaringChain
    from nim_groth16_amd.zkey_types import packCoeffs
    for log2n in (16, 18):
        m = (1 << log2n) - 2
        r1cs, wit = squaringChain(m, seed=4)
        rng = SplitMix64(5)
        zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
        pk = loadProvingKey(zk, ctx)
        wb = _bytes(wit)
        mask = Mask(rng.fr(), rng.fr())
        pr = generateProofWithMask(0, False, zk, Witness("bn128", m + 2, wb), mask, ctx, pkey=pk)
        assert verifyProof(extractVKey(zk), pr, ctx)                          # testProver.nim:65-73 at scale
        if log2n == 16:   # the five MSMs and the quotient recomputed by the C oracle, then the O(1) mask algebra
            pts, hdr = zk.pPoints, zk.header
            Az, Bz, Cz = orc.build_abc(packCoeffs(zk.coeffs), wb, log2n)
            qs = orc.quotient_snarkjs(Az, Bz, Cz, log2n)
            it = iter([o.g1_from_bytes(orc.msm(1, wb, pts.pointsA1)), o.g1_from_bytes(orc.msm(1, wb, pts.pointsB1)),
                       o.g2_from_bytes(orc.msm(2, wb, pts.pointsB2)), o.g1_from_bytes(orc.msm(1, qs, pts.pointsH1)),
                       o.g1_from_bytes(orc.msm(1, wb[32 * (hdr.npubs + 1):], pts.pointsC1))])
            oz = o.ZKey()
            oz.flavour, oz.nvars, oz.npubs, oz.domainSize = o.SNARKJS, hdr.nvars, hdr.npubs, hdr.domainSize
            sp = zk.specPoints
            oz.alpha1, oz.beta1, oz.delta1 = (o.g1_from_bytes(x) for x in (sp.alpha1, sp.beta1, sp.delta1))
            oz.beta2, oz.gamma2, oz.delta2 = (o.g2_from_bytes(x) for x in (sp.beta2, sp.gamma2, sp.delta2))
            oz.pointsA1 = oz.pointsB1 = oz.pointsB2 = [None] * hdr.nvars
            oz.pointsC1, oz.pointsH1, oz.coeffs = [None] * (hdr.nvars - hdr.npubs - 1), [None] * hdr.domainSize, []
            ref = o.generate_proof_with_mask(oz, wit, mask.r, mask.s, msm_g1=lambda c_, p_: next(it),
                                             msm_g2=lambda c_, p_: next(it), quotient=lambda *a: [0] * hdr.domainSize)
            assert (o.g1_from_bytes(pr.pi_a), o.g2_from_bytes(pr.pi_b), o.g1_from_bytes(pr.pi_c)) == \
                (ref.pi_a, ref.pi_b, ref.pi_c)
        pk.destroy()
