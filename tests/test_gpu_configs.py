"""The BASELINE.json configurations the green run of round 2 never touched, inside `-m gpu`:
  config 4    2^22-constraint circuit, MSMs point-sharded over 8 ranks, shard by shard on the one GPU of the box
  config 2-ii circom-like scalars (40 % zero, 30 % one, 10 % < 2^16, 20 % uniform) at 2^16 and 2^20, G1 and G2
  config 5's shape: a full proof on a circuit whose witness is ~70 % zeros / ones (the heavy-bucket path), and keys
              with (0,0) points for wires absent from B
  and the full-size NTT / quotient held to EVERY output of the C oracle (not to a few indices).
Everything on the GPU goes through the C ABI; the oracle is the checker."""
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


def _toxic(seed=5):
    from nim_groth16_amd.fake_setup import ToxicWaste
    from nim_groth16_amd.synthetic import SplitMix64
    rng = SplitMix64(seed)
    return ToxicWaste(*[rng.fr() for _ in range(5)]), rng


# ---- config 4 --------------------------------------------------------------------------------------------------
def test_config4_2p22_sharded_over_8_ranks_by_hand(ctx, orc):
    """groth16/prover.nim:279-302 at fake_setup.nim:203-206's domain rule, n = 2^22, with the chunk rule of
    msm.nim:105-115 over G = 8 ranks: every rank's g16_prove_partials_begin (its coset pipelines + its four witness
    MSMs), the [h_lo, h_hi) slices handed over by hand in place of the three scatters, every rank's _end,
    g16_prove_combine of the 8 records.  The result must equal the unsharded key's proof, be accepted by the GPU
    verifier and equal the C oracle's proof bit for bit."""
    import torch
    from nim_groth16_amd import Context, Mask, Proof, extractVKey, loadProvingKey, verifyProof
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.distributed import quotientTaskOwner, shardRange
    from nim_groth16_amd.fake_setup import fakeCircuitSetup
    from nim_groth16_amd.synthetic import squaringChain
    from tests.parity import check_gpu_proof
    log2n, G = 22, 8
    n = 1 << log2n
    r1cs, wit = squaringChain(n - 2, seed=4)
    tox, rng = _toxic()
    zk = fakeCircuitSetup(r1cs, tox, 1, ctx)
    assert zk.header.domainSize == n
    wb = _bytes(wit)
    mask = Mask(rng.fr(), rng.fr())
    rb, sb = F.frToMontBytes(mask.r), F.frToMontBytes(mask.s)
    d_w = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda()
    pk = loadProvingKey(zk, ctx)
    whole = pk.prove(d_w.data_ptr(), mont=True, r=rb, s=sb, device=True)
    pk.destroy()
    # one context per rank: a context carries ONE pending begin (include/g16hip.h)
    ranks = [Context(0) for _ in range(G)]
    keys, vecs = [], {}
    try:
        for rank in range(G):
            k = loadProvingKey(zk, ranks[rank], shard_index=rank, shard_count=G)
            keys.append(k)
            owned = [v for v in range(3) if quotientTaskOwner(v, G) == rank]
            out = torch.empty(max(1, len(owned)) * n * 32, dtype=torch.uint8, device="cuda")
            k.prove_partials_begin(d_w.data_ptr(), sum(1 << v for v in owned), out.data_ptr() if owned else None,
                                   device=True)
            for i, v in enumerate(owned):
                vecs[v] = out[32 * n * i: 32 * n * (i + 1)]
        recs = b""
        for rank in range(G):
            lo, hi = shardRange(n, rank, G)
            sl = [vecs[v][32 * lo: 32 * hi].contiguous() for v in range(3)]
            torch.cuda.synchronize()
            recs += keys[rank].prove_partials_end(*[s_.data_ptr() for s_ in sl])
        sharded = keys[0].prove_combine(recs, G, rb, sb)
    finally:
        for k in keys:
            k.destroy()
        for c in ranks:
            c.close()
    assert sharded == whole, "the proof combined from 8 shard records differs from the unsharded proof"
    # the same configuration through the C-ABI device group (g16_group_*: eight members, one host thread each inside the
    # library, slices moved by device copies) -- what a Nim host calls instead of driving the ranks itself
    from nim_groth16_amd import DeviceGroup, loadGroupKey
    grp = DeviceGroup([0] * G)
    gk = loadGroupKey(zk, grp)
    try:
        assert gk.prove(wb, r=rb, s=sb) == whole, "g16_group_prove over 8 members differs from the unsharded proof"
    finally:
        gk.destroy()
        grp.close()
    pio = wb[:32 * (zk.header.npubs + 1)]
    assert verifyProof(extractVKey(zk), Proof(pio, *sharded), ctx)
    check_gpu_proof(orc, zk, wit, wb, mask.r, mask.s, sharded, ctx)


# ---- config 2 (ii) -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("group,log2n", [(1, 16), (2, 16), (1, 20), (2, 20)])
def test_registered_msm_circom_like_scalars(ctx, orc, group, log2n):
    """msmMultiThreadedG1/G2 (msm.nim:89-158) on registered sets with the scalar mix of a circom witness: 30 % of all
    scalars land in ONE bucket, which becomes ~n/L accumulate segments and one msm_heavy LDS tree (msm.cuh).  Closed
    form: P_i = k_i G, expectation (sum s_i k_i) G from the oracle's own scalar multiplication."""
    n = 1 << log2n
    ks = _stream(1, n)
    sc = I.circom_like_scalars(n, seed=3)
    assert 0.38 < sum(1 for s in sc if s == 0) / n < 0.42 and 0.28 < sum(1 for s in sc if s == 1) / n < 0.32
    pts = ctx.fixed_base(group, _bytes(ks))
    psz = 64 * group
    sample = [0, 1, n // 3, n - 1]
    assert b"".join(pts[psz * i:psz * (i + 1)] for i in sample) == orc.fixed_base(group, _bytes([ks[i] for i in sample]))
    e = sum(s * k for s, k in zip(sc, ks)) % R
    gen = o.g1_to_bytes(o.GEN1) if group == 1 else o.g2_to_bytes(o.GEN2)
    exp = orc.mul(group, _bytes([e]), gen)
    h = ctx.register_points(group, pts, n)
    try:
        assert ctx.msm_points(h, _bytes(sc)) == exp                           # Montgomery scalars (Nim seq[Fr])
        assert ctx.msm_points(h, I.fr_std_bytes(sc), mont=False) == exp        # the .wtns layout
    finally:
        h.release()
    if log2n == 16:   # BASELINE config 2 proper: the one-shot G1 / G2 MSM of 2^16 pairs, and the C oracle's Pippenger
        assert ctx.msm(group, _bytes(sc), pts, n) == exp
        assert orc.msm(group, _bytes(sc), pts) == exp


# ---- config 5's shape ---------------------------------------------------------------------------------------------
def _prove_mixed(ctx, orc, log2n, **kw):
    from nim_groth16_amd import Mask, Witness, extractVKey, generateProofWithMask, loadProvingKey, verifyProof
    from nim_groth16_amd.fake_setup import fakeCircuitSetup
    from nim_groth16_amd.synthetic import mixedCircuit
    from tests.parity import check_gpu_proof
    m = (1 << log2n) - 2
    r1cs, wit = mixedCircuit(m, seed=4, **kw)
    tox, rng = _toxic()
    zk = fakeCircuitSetup(r1cs, tox, 1, ctx)
    pk = loadProvingKey(zk, ctx)
    try:
        wb = _bytes(wit)
        mask = Mask(rng.fr(), rng.fr())
        pr = generateProofWithMask(0, False, zk, Witness("bn128", m + 2, wb), mask, ctx, pkey=pk)
        assert verifyProof(extractVKey(zk), pr, ctx)
        check_gpu_proof(orc, zk, wit, wb, mask.r, mask.s, (pr.pi_a, pr.pi_b, pr.pi_c), ctx)
        std = generateProofWithMask(0, False, zk, Witness("bn128", m + 2, I.fr_std_bytes(wit), std=True), mask, ctx,
                                    pkey=pk)
        assert (std.pi_a, std.pi_b, std.pi_c) == (pr.pi_a, pr.pi_b, pr.pi_c)
        inf = pk.inf_counts()
    finally:
        pk.destroy()
    return zk, wit, inf


def test_full_proof_2p18_circom_like_witness_bit_exact(ctx, orc):
    """a 2^18-constraint circuit whose witness is 40 % zeros and 30 % ones (booleanity rows b*b = b next to the
    squaring chain): the four witness MSMs share one sort in which a single bucket holds ~30 % of all entries
    (msm_accum extra segments -> msm_heavy).  Bit-exact vs the C oracle, both witness encodings."""
    zk, wit, inf = _prove_mixed(ctx, orc, 18)
    assert not inf["compact_A"] and not inf["compact_B"]       # every wire occurs in A and in B: dense sets, ONE sort
    nz = sum(1 for w in wit if w == 0) / len(wit)
    no = sum(1 for w in wit if w == 1) / len(wit)
    assert nz + no >= 0.6 and nz > 0.35 and no > 0.25


@pytest.mark.parametrize("lin_pct,log2n", [(50, 14), (90, 16)])
def test_full_proof_with_infinity_points_in_B(ctx, orc, lin_pct, log2n):
    """snarkjs keys carry (0,0) for every wire absent from B (curves.nim:95-107 accepts them, msm.nim:128-158 sums
    over them): here 50 % / 90 % of pointsB1 and pointsB2 are the point at infinity.  Bit-exact vs the C oracle."""
    zk, _, inf = _prove_mixed(ctx, orc, log2n, zero_pct=0, one_pct=0, lin_pct=lin_pct)
    b2 = zk.pPoints.pointsB2
    ninf = sum(1 for i in range(0, len(b2), 128) if b2[i:i + 128] == bytes(128))
    b1 = zk.pPoints.pointsB1
    assert ninf == sum(1 for i in range(0, len(b1), 64) if b1[i:i + 64] == bytes(64))
    frac = ninf / zk.header.nvars
    assert abs(frac - lin_pct / 100) < 0.08, frac
    # the library saw the same points, and gave B1 / B2 their own entry lists without them (g16_pkey_inf_counts)
    assert inf["B1"] == inf["B2"] == inf["B1_and_B2"] == ninf and inf["compact_B"] and not inf["compact_A"]


@pytest.mark.parametrize("group", [1, 2])
def test_registered_msm_with_mostly_infinity_points(ctx, orc, group):
    """a registered set with 60 % (0,0) points runs on compacted entry lists (g16_points_inf_count); result == the
    oracle's MSM over the same arrays, which adds the infinities the long way (msm.nim:162-198 semantics)"""
    n = 1 << 13
    ks, sc = _stream(1, n), _stream(2, n)
    psz = 64 * group
    pts = bytearray(ctx.fixed_base(group, _bytes(ks)))
    dead = [i for i in range(n) if (i * 2654435761) % 100 < 60]
    for i in dead:
        pts[psz * i: psz * (i + 1)] = bytes(psz)
    pts = bytes(pts)
    h = ctx.register_points(group, pts, n)
    try:
        assert h.inf_count() == len(dead)
        got = ctx.msm_points(h, _bytes(sc))
    finally:
        h.release()
    assert got == orc.msm(group, _bytes(sc), pts)
    assert got == ctx.msm(group, _bytes(sc), pts, n)


# ---- full-size NTT and quotient against every output of the C oracle ----------------------------------------------
def test_ntt_2p20_all_outputs_vs_c_oracle(ctx, orc):
    """forwardNTT / inverseNTT (ntt.nim:55-77, 139-161) at the benchmark size, all 2^20 outputs of both directions"""
    log2n, n = 20, 1 << 20
    xb = _bytes(_stream(31, n))
    assert ctx.ntt(xb, log2n, False) == orc.ntt(xb, log2n, inverse=False)
    assert ctx.ntt(xb, log2n, True) == orc.ntt(xb, log2n, inverse=True)


@pytest.mark.parametrize("flavour", [1, 0])
def test_quotient_2p20_all_outputs_vs_c_oracle(ctx, orc, flavour):
    """computeSnarkjsScalarCoeffs (prover.nim:158-181) / computeQuotientPointwise (:118-148) at 2^20, every output"""
    log2n, n = 20, 1 << 20
    ab, bb, cb = (_bytes(_stream(s, n)) for s in (41, 42, 43))
    got = ctx.quotient(ab, bb, cb, log2n, flavour)
    want = orc.quotient_snarkjs(ab, bb, cb, log2n) if flavour == 1 else orc.quotient_jensgroth(ab, bb, cb, log2n)
    assert got == want
