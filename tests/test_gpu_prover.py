"""GPU parity for the prover-level path (through the C ABI): fixed-base setup, buildABC, quotient (both
flavours) and the full proof, bit-exact against the oracle; plus the reference's own check -- the proof
verifies (tests/groth16/testProver.nim:59-73) -- on the reference's toy circuit and on a synthetic chain."""
import pytest

from oracle import bn254_ref as o
from tests import inputs as I

pytestmark = pytest.mark.gpu


def _zkey_to_oracle(zk):
    """product ZKey (bytes) -> oracle ZKey (ints), so the oracle can verify / re-prove it."""
    z = o.ZKey()
    h = zk.header
    z.flavour = o.SNARKJS if h.flavour == 1 else o.JENS_GROTH
    z.nvars, z.npubs, z.domainSize, z.logDomainSize = h.nvars, h.npubs, h.domainSize, h.logDomainSize
    g1s = lambda b: [o.g1_from_bytes(b[i:i + 64]) for i in range(0, len(b), 64)]      # noqa: E731
    g2s = lambda b: [o.g2_from_bytes(b[i:i + 128]) for i in range(0, len(b), 128)]    # noqa: E731
    sp = zk.specPoints
    z.alpha1, z.beta1, z.delta1 = g1s(sp.alpha1)[0], g1s(sp.beta1)[0], g1s(sp.delta1)[0]
    z.beta2, z.gamma2, z.delta2 = g2s(sp.beta2)[0], g2s(sp.gamma2)[0], g2s(sp.delta2)[0]
    z.pointsIC = g1s(zk.pointsIC)
    p = zk.pPoints
    z.pointsA1, z.pointsB1, z.pointsB2 = g1s(p.pointsA1), g1s(p.pointsB1), g2s(p.pointsB2)
    z.pointsC1, z.pointsH1 = g1s(p.pointsC1), g1s(p.pointsH1)
    z.coeffs = [(m, r, c, o.fr_from_mont_bytes(v)) for (m, r, c, v) in zk.coeffs]
    return z


def _toxic(seed):
    rng = o.SplitMix64(seed)
    return [rng.fr() for _ in range(5)]


@pytest.mark.parametrize("group", [1, 2])
def test_fixed_base_vs_oracle(ctx, orc, group):
    ks = I.uniform_scalars(300, 11) + [0, 1, o.R - 1, 255, 256, 1 << 248]
    sb = I.fr_mont_bytes(ks)
    assert ctx.fixed_base(group, sb) == orc.fixed_base(group, sb)
    assert ctx.fixed_base(group, I.fr_std_bytes(ks), mont=False) == orc.fixed_base(group, sb)
    assert ctx.fixed_base(group, b"") == b""


@pytest.mark.parametrize("flavour", [0, 1])
def test_toy_circuit_setup_prove_verify_bit_exact(ctx, flavour):
    """the reference's fixture (tests/groth16/testProver.nim:17-73), both flavours, masks (0,0) and random"""
    from nim_groth16_amd import Mask, Witness, generateProofWithMask, generateProofWithTrivialMask
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    a, b, g, d, t = _toxic(5)
    toy = o.toy_r1cs()
    zk = fakeCircuitSetup(R1CS(8, 1, 1, 3, toy.constraints), ToxicWaste(a, b, g, d, t), flavour, ctx)
    oz = o.fake_circuit_setup(toy, o.ToxicWaste(a, b, g, d, t), o.SNARKJS if flavour else o.JENS_GROTH)
    z2 = _zkey_to_oracle(zk)
    for f in ("alpha1", "beta1", "delta1", "beta2", "gamma2", "delta2", "pointsIC", "pointsA1", "pointsB1",
              "pointsB2", "pointsC1", "pointsH1", "coeffs"):
        assert getattr(z2, f) == getattr(oz, f), f
    wt = Witness("bn128", 8, I.fr_mont_bytes(o.TOY_WITNESS))
    rng = o.SplitMix64(6)
    for (r, s) in ((0, 0), (rng.fr(), rng.fr()), (1, o.R - 1)):
        pr = generateProofWithMask(0, False, zk, wt, Mask(r, s), ctx) if (r or s) else \
            generateProofWithTrivialMask(0, False, zk, wt, ctx)
        ref = o.generate_proof_with_mask(oz, o.TOY_WITNESS, r, s)
        assert o.g1_from_bytes(pr.pi_a) == ref.pi_a
        assert o.g2_from_bytes(pr.pi_b) == ref.pi_b
        assert o.g1_from_bytes(pr.pi_c) == ref.pi_c
        assert I.fr_from_mont(pr.publicIO) == ref.publicIO == [1, 2023, 1022]   # prover.nim:238-240
        assert o.verify_proof(oz, ref)                      # verifier.nim:31-52


@pytest.mark.parametrize("log2n", [1, 3, 8, 11, 14, 17])
@pytest.mark.parametrize("flavour", [0, 1])
def test_quotient_vs_oracle(ctx, orc, log2n, flavour):
    n = 1 << log2n
    A, B, C = (I.uniform_scalars(n, s) for s in (21, 22, 23))
    got = ctx.quotient(I.fr_mont_bytes(A), I.fr_mont_bytes(B), I.fr_mont_bytes(C), log2n, flavour)
    ab, bb, cb = I.fr_mont_bytes(A), I.fr_mont_bytes(B), I.fr_mont_bytes(C)
    # every (flavour, size) pair is held to the C oracle; the small ones also to the Python transliteration
    assert got == (orc.quotient_snarkjs(ab, bb, cb, log2n) if flavour == 1 else
                   orc.quotient_jensgroth(ab, bb, cb, log2n))
    if log2n <= 8:
        exp = o.compute_snarkjs_scalar_coeffs(A, B, C) if flavour else o.compute_quotient_pointwise(A, B, C)
        assert I.fr_from_mont(got) == exp


def test_synthetic_chain_prove_matches_oracle_and_verifies(ctx, orc):
    """squaring chain, m = 2^9 - 2 constraints (domain 2^9): buildABC, proof bytes and verification"""
    from nim_groth16_amd import Mask, Witness, generateProofWithMask, loadProvingKey
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import squaringChain
    m = (1 << 9) - 2
    r1cs, wit = squaringChain(m, seed=4)
    a, b, g, d, t = _toxic(5)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(a, b, g, d, t), 1, ctx)
    assert zk.header.domainSize == 1 << 9 and zk.header.nvars == m + 2
    # oracle setup with the C oracle's fixed-base multiplier
    bg1 = lambda ks: [o.g1_from_bytes(x) for x in _chunks(orc.fixed_base(1, I.fr_mont_bytes(ks)), 64)]    # noqa: E731
    bg2 = lambda ks: [o.g2_from_bytes(x) for x in _chunks(orc.fixed_base(2, I.fr_mont_bytes(ks)), 128)]   # noqa: E731
    oz = o.fake_circuit_setup(o.R1CS(r1cs.nWires, 1, 0, 1, r1cs.constraints), o.ToxicWaste(a, b, g, d, t),
                              o.SNARKJS, bg1, bg2)
    z2 = _zkey_to_oracle(zk)
    assert z2.pointsA1 == oz.pointsA1 and z2.pointsB2 == oz.pointsB2 and z2.pointsC1 == oz.pointsC1
    assert z2.pointsH1 == oz.pointsH1 and z2.coeffs == oz.coeffs
    pk = loadProvingKey(zk, ctx)
    wb = I.fr_mont_bytes(wit)
    Az, Bz, Cz = pk.build_abc(wb)
    eA, eB, eC = o.build_abc(oz.coeffs, oz.domainSize, wit)
    assert I.fr_from_mont(Az) == eA and I.fr_from_mont(Bz) == eB and I.fr_from_mont(Cz) == eC
    rng = o.SplitMix64(6)
    r, s = rng.fr(), rng.fr()
    pr = generateProofWithMask(0, False, zk, Witness("bn128", m + 2, wb), Mask(r, s), ctx, pkey=pk)
    msm1 = lambda cs, ps: o.g1_from_bytes(orc.msm(1, I.fr_mont_bytes(cs), b"".join(o.g1_to_bytes(p) for p in ps)))   # noqa: E731
    msm2 = lambda cs, ps: o.g2_from_bytes(orc.msm(2, I.fr_mont_bytes(cs), b"".join(o.g2_to_bytes(p) for p in ps)))   # noqa: E731
    ref = o.generate_proof_with_mask(oz, wit, r, s, msm_g1=msm1, msm_g2=msm2)
    assert (o.g1_from_bytes(pr.pi_a), o.g2_from_bytes(pr.pi_b), o.g1_from_bytes(pr.pi_c)) == \
        (ref.pi_a, ref.pi_b, ref.pi_c)
    assert o.verify_proof(oz, ref)
    # std-form witness (raw .wtns layout) gives the same proof
    pa, pb, pc = pk.prove(I.fr_std_bytes(wit), mont=False, r=o.fr_to_mont_bytes(r), s=o.fr_to_mont_bytes(s))
    assert (pa, pb, pc) == (pr.pi_a, pr.pi_b, pr.pi_c)
    pk.destroy()


def _chunks(b, k):
    return [b[i:i + k] for i in range(0, len(b), k)]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_proof_single_gpu(ctx, world):
    """config 4 shape on one device: `world` sharded keys (msm.nim:105-115 ranges), their 768-byte partial
    records concatenated in rank order, g16_prove_combine -> the same proof as the unsharded key."""
    from nim_groth16_amd import loadProvingKey
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import squaringChain
    m = (1 << 7) - 2
    r1cs, wit = squaringChain(m, seed=9)
    a, b, g, d, t = _toxic(15)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(a, b, g, d, t), 1, ctx)
    wb = I.fr_mont_bytes(wit)
    rng = o.SplitMix64(16)
    r, s = o.fr_to_mont_bytes(rng.fr()), o.fr_to_mont_bytes(rng.fr())
    full = loadProvingKey(zk, ctx)
    want = full.prove(wb, r=r, s=s)
    keys = [loadProvingKey(zk, ctx, shard_index=k, shard_count=world) for k in range(world)]
    recs = b"".join(k.prove_partials(wb) for k in keys)
    assert len(recs) == 768 * world
    for k in keys:
        assert k.prove_combine(recs, world, r, s) == want
    with pytest.raises(Exception):
        keys[0].prove(wb)                      # a sharded key cannot prove alone
    for k in keys + [full]:
        k.destroy()


@pytest.mark.parametrize("world,log2n", [(1, 9), (2, 11), (3, 11), (8, 12), (5, 4)])
def test_sharded_proof_with_task_parallel_quotient_single_gpu(ctx, orc, world, log2n):
    """g16_prove_partials_begin / _end on one device, the exchange done by hand: `world` contexts stand for the ranks;
    rank v mod world computes coset vector v (prover.nim:167-169's three tasks), every rank gets its [h_lo, h_hi)
    slices, forms A1*B1 - C1 there and runs its share of the H MSM.  The combined proof must equal the unsharded
    key's, and each coset vector the oracle's shiftEvalDomain (prover.nim:109-113)."""
    import torch
    from nim_groth16_amd import Context, loadProvingKey
    from nim_groth16_amd.distributed import quotientTaskOwner, shardRange
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import squaringChain
    n = 1 << log2n
    m = n - 2
    r1cs, wit = squaringChain(m, seed=9)
    a, b, g, d, t = _toxic(15)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(a, b, g, d, t), 1, ctx)
    wb = I.fr_mont_bytes(wit)
    rng = o.SplitMix64(16)
    r, s = o.fr_to_mont_bytes(rng.fr()), o.fr_to_mont_bytes(rng.fr())
    full = loadProvingKey(zk, ctx)
    want = full.prove(wb, r=r, s=s)
    ctxs = [Context(0) for _ in range(world)]
    keys = [loadProvingKey(zk, ctxs[k], shard_index=k, shard_count=world) for k in range(world)]
    vecs = {}
    for k in range(world):
        owned = [v for v in range(3) if quotientTaskOwner(v, world) == k]
        out = torch.empty(max(1, len(owned)) * n * 32, dtype=torch.uint8, device="cuda")
        keys[k].prove_partials_begin(wb, sum(1 << v for v in owned), out.data_ptr() if owned else None)
        for i, v in enumerate(owned):
            vecs[v] = out[32 * n * i: 32 * n * (i + 1)].clone()
    torch.cuda.synchronize()
    # the three coset vectors against the C oracle: quotient_snarkjs = A1*B1 - C1, so check through it
    Az, Bz, Cz = full.build_abc(wb)
    qs = orc.quotient_snarkjs(Az, Bz, Cz, log2n)
    A1, B1, C1 = (I.fr_from_mont(bytes(vecs[v].cpu().numpy())) for v in range(3))
    assert [(x * y - z) % o.R for x, y, z in zip(A1, B1, C1)] == I.fr_from_mont(qs)
    if log2n <= 9:
        assert A1 == o.shift_eval_domain(I.fr_from_mont(Az), o.Domain(n), o.Domain(2 * n).domainGen)
    recs = b""
    for k in range(world):
        lo, hi = shardRange(n, k, world)
        sl = [vecs[v][32 * lo: 32 * hi].contiguous() for v in range(3)]
        torch.cuda.synchronize()
        recs += keys[k].prove_partials_end(*[x.data_ptr() if hi > lo else None for x in sl])
    assert len(recs) == 768 * world
    assert keys[0].prove_combine(recs, world, r, s) == want
    # protocol errors come back as G16_EINVAL, not as a crash
    from nim_groth16_amd import G16Error
    with pytest.raises(G16Error):
        keys[0].prove_partials_end(None, None, None)          # no begin outstanding
    for k in keys + [full]:
        k.destroy()
    for c in ctxs:
        c.close()


def test_files_to_proof_json_end_to_end(ctx, tmp_path):
    """snarkjs-style pipeline (groth16/example/prove.sh:52-59 without circom/snarkjs): .zkey + .wtns files ->
    parse -> GPU prove (raw standard-form witness bytes) -> proof.json / public.json -> verified by the oracle"""
    import json
    from nim_groth16_amd import Mask, generateProofWithMask
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.files import exportProof, exportPublicIO, parseWitness, parseZKey, writeWitness, writeZKey
    a, b, g, d, t = _toxic(25)
    toy = o.toy_r1cs()
    zk0 = fakeCircuitSetup(R1CS(8, 1, 1, 3, toy.constraints), ToxicWaste(a, b, g, d, t), 1, ctx)
    zpath, wpath = str(tmp_path / "c.zkey"), str(tmp_path / "c.wtns")
    writeZKey(zpath, zk0)
    writeWitness(wpath, o.TOY_WITNESS)
    zk, wt = parseZKey(zpath), parseWitness(wpath)
    assert zk == zk0 and wt.std
    rng = o.SplitMix64(26)
    r, s = rng.fr(), rng.fr()
    pr = generateProofWithMask(0, False, zk, wt, Mask(r, s), ctx)
    pj, ij = str(tmp_path / "proof.json"), str(tmp_path / "public.json")
    exportProof(pj, pr)
    exportPublicIO(ij, pr)
    dj = json.load(open(pj))
    oz = o.fake_circuit_setup(toy, o.ToxicWaste(a, b, g, d, t), o.SNARKJS)
    ref = o.generate_proof_with_mask(oz, o.TOY_WITNESS, r, s)
    got = o.Proof([1] + [int(x) for x in json.load(open(ij))],
                  (int(dj["pi_a"][0]), int(dj["pi_a"][1])),
                  ((int(dj["pi_b"][0][0]), int(dj["pi_b"][0][1])), (int(dj["pi_b"][1][0]), int(dj["pi_b"][1][1]))),
                  (int(dj["pi_c"][0]), int(dj["pi_c"][1])))
    assert (got.pi_a, got.pi_b, got.pi_c, got.publicIO) == (ref.pi_a, ref.pi_b, ref.pi_c, ref.publicIO)
    assert o.verify_proof(oz, got)


def _oracle_setup(r1cs_o, tox, flavour, orc):
    bg1 = lambda ks: [o.g1_from_bytes(x) for x in _chunks(orc.fixed_base(1, I.fr_mont_bytes(ks)), 64)]    # noqa: E731
    bg2 = lambda ks: [o.g2_from_bytes(x) for x in _chunks(orc.fixed_base(2, I.fr_mont_bytes(ks)), 128)]   # noqa: E731
    return o.fake_circuit_setup(r1cs_o, tox, flavour, bg1, bg2)


def _oracle_prove(oz, wit, r, s, orc):
    msm1 = lambda cs, ps: o.g1_from_bytes(orc.msm(1, I.fr_mont_bytes(cs), b"".join(map(o.g1_to_bytes, ps)))) if cs else o.INF_G1   # noqa: E731
    msm2 = lambda cs, ps: o.g2_from_bytes(orc.msm(2, I.fr_mont_bytes(cs), b"".join(map(o.g2_to_bytes, ps)))) if cs else o.INF_G2   # noqa: E731
    return o.generate_proof_with_mask(oz, wit, r, s, msm_g1=msm1, msm_g2=msm2)


@pytest.mark.parametrize("case", ["no_public_signals", "all_wires_public", "tiny_domain", "jensgroth_mid",
                                  "sparse_zero_witness"])
def test_prover_shape_edge_cases(ctx, orc, case):
    """shapes at the edges of generateProofWithMask's asserts (prover.nim:236, 270-276): npubs = 0; an empty
    pointsC1 (nvars = npubs+1); the smallest domains; the JensGroth flavour beyond the toy size (7 NTTs,
    prover.nim:118-148); a witness that is almost all zeros (empty buckets everywhere)."""
    from nim_groth16_amd import Mask, Witness, generateProofWithMask, loadProvingKey
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import squaringChain
    a, b, g, d, t = _toxic(31)
    flavour = 1
    if case == "no_public_signals":          # x*x = y with nothing public: npubs = 0 -> domain from 1+0+1
        cons = [([(1, 1)], [(1, 1)], [(2, 1)])]
        r1, wit = R1CS(3, 0, 0, 1, cons), [1, 5, 25]
    elif case == "all_wires_public":         # nvars = npubs + 1: pointsC1 is empty, zs = []
        cons = [([(1, 1)], [(1, 1)], [(2, 1)])]
        r1, wit = R1CS(3, 1, 1, 0, cons), [1, 7, 49]
    elif case == "tiny_domain":              # one constraint, one public output -> domain 4
        cons = [([(2, 1)], [(2, 1)], [(1, 1)])]
        r1, wit = R1CS(3, 1, 0, 1, cons), [1, 81, 9]
    elif case == "jensgroth_mid":
        r1, wit = squaringChain((1 << 7) - 2, seed=8)
        flavour = 0
    else:                                    # chain started at 0 with k_i = 0 is impossible (k random); zero most wires instead
        m = (1 << 6) - 2
        cons = [([(i + 2, 1)], [(i + 2, 1)], [((i + 3) if i + 1 < m else 1, 1)]) for i in range(m)]
        r1, wit = R1CS(m + 2, 1, 0, 1, cons), [1] + [0] * (m + 1)
    zk = fakeCircuitSetup(r1, ToxicWaste(a, b, g, d, t), flavour, ctx)
    oz = _oracle_setup(o.R1CS(r1.nWires, r1.nPubOut, r1.nPubIn, r1.nPrivIn, r1.constraints),
                       o.ToxicWaste(a, b, g, d, t), o.SNARKJS if flavour else o.JENS_GROTH, orc)
    # the witness satisfies the circuit
    for (A, B, C) in r1.constraints:
        ev = lambda lc: sum(v * wit[w] for w, v in lc) % o.R        # noqa: E731
        assert ev(A) * ev(B) % o.R == ev(C)
    pk = loadProvingKey(zk, ctx)
    rng = o.SplitMix64(32)
    for (r, s) in ((0, 0), (rng.fr(), rng.fr())):
        pr = generateProofWithMask(0, False, zk, Witness("bn128", r1.nWires, I.fr_mont_bytes(wit)), Mask(r, s), ctx, pkey=pk)
        ref = _oracle_prove(oz, wit, r, s, orc)
        assert (o.g1_from_bytes(pr.pi_a), o.g2_from_bytes(pr.pi_b), o.g1_from_bytes(pr.pi_c)) == \
            (ref.pi_a, ref.pi_b, ref.pi_c), case
        assert o.verify_proof(oz, ref), case
    pk.destroy()


def test_prover_rejects_bad_shapes(ctx):
    """error convention at the boundary: bad arguments come back as G16_EINVAL / AssertionError, never a crash"""
    from nim_groth16_amd import G16Error, Mask, Witness, generateProofWithMask
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.zkey_types import MatrixC
    a, b, g, d, t = _toxic(41)
    toy = o.toy_r1cs()
    zk = fakeCircuitSetup(R1CS(8, 1, 1, 3, toy.constraints), ToxicWaste(a, b, g, d, t), 1, ctx)
    with pytest.raises(AssertionError):       # prover.nim:236 "wrong witness length"
        generateProofWithMask(0, False, zk, Witness("bn128", 7, I.fr_mont_bytes(o.TOY_WITNESS[:7])), Mask(0, 0), ctx)
    with pytest.raises(AssertionError):       # prover.nim:224 curve mismatch
        generateProofWithMask(0, False, zk, Witness("bls12", 8, I.fr_mont_bytes(o.TOY_WITNESS)), Mask(0, 0), ctx)
    import copy
    bad = copy.deepcopy(zk)
    bad.coeffs = list(bad.coeffs) + [(MatrixC, 0, 0, o.fr_to_mont_bytes(1))]
    from nim_groth16_amd import loadProvingKey
    with pytest.raises(G16Error) as e:        # MatrixC entries make buildABC raise (prover.nim:67)
        loadProvingKey(bad, ctx)
    assert e.value.code == -1


def test_zkey_point_check_on_gpu(ctx, tmp_path):
    """mkG1 / mkG2 on-curve asserts of the reference's loaders (curves.nim:95-107) as a GPU pass"""
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.files import parseZKey, writeZKey
    a, b, g, d, t = _toxic(51)
    zk = fakeCircuitSetup(R1CS(8, 1, 1, 3, o.toy_r1cs().constraints), ToxicWaste(a, b, g, d, t), 1, ctx)
    path = str(tmp_path / "k.zkey")
    writeZKey(path, zk)
    assert parseZKey(path, check=True, ctx=ctx) == zk
    pts = bytearray(zk.pPoints.pointsB2)
    assert ctx.points_check(2, bytes(pts)) is None
    pts[128 * 3 + 5] ^= 1                                    # corrupt point 3
    assert ctx.points_check(2, bytes(pts)) == 3
    g1 = bytearray(zk.pPoints.pointsA1)
    assert ctx.points_check(1, bytes(g1) + bytes(64)) is None   # (0,0) = infinity is accepted (curves.nim:96-98)
    g1[64 * 5 + 40] ^= 0x80
    assert ctx.points_check(1, bytes(g1)) == 5
    assert ctx.points_check(1, b"") is None
    import dataclasses
    bad = dataclasses.replace(zk, pPoints=dataclasses.replace(zk.pPoints, pointsA1=bytes(g1)))
    writeZKey(path, bad)
    with pytest.raises(AssertionError):
        parseZKey(path, check=True, ctx=ctx)


def test_named_path_functions_mirror_the_reference(ctx):
    """buildABC / computeSnarkjsScalarCoeffs / computeQuotientPointwise / polyForwardNTT / polyInverseNTT under the
    reference's names (prover.nim:56-181, poly.nim:255-268), on the toy circuit, against the oracle"""
    from nim_groth16_amd import (ABC, buildABC, computeQuotientPointwise, computeSnarkjsScalarCoeffs, createDomain,
                                 polyForwardNTT, polyInverseNTT)
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    a, b, g, d, t = _toxic(5)
    toy = o.toy_r1cs()
    zk = fakeCircuitSetup(R1CS(8, 1, 1, 3, toy.constraints), ToxicWaste(a, b, g, d, t), 1, ctx)
    abc = buildABC(zk, I.fr_mont_bytes(o.TOY_WITNESS), ctx)
    Az, Bz, Cz = o.build_abc(o.r1cs_to_coeffs(toy), 8, o.TOY_WITNESS)
    assert isinstance(abc, ABC)
    assert (I.fr_from_mont(abc.valuesAz), I.fr_from_mont(abc.valuesBz), I.fr_from_mont(abc.valuesCz)) == (Az, Bz, Cz)
    assert I.fr_from_mont(computeSnarkjsScalarCoeffs(0, abc, ctx)) == o.compute_snarkjs_scalar_coeffs(Az, Bz, Cz)
    assert I.fr_from_mont(computeQuotientPointwise(0, abc, ctx)) == o.compute_quotient_pointwise(Az, Bz, Cz)
    D = createDomain(16)
    coeffs = I.uniform_scalars(11, 41)                       # shorter than the domain: padded (poly.nim:257-259)
    ys = polyForwardNTT(I.fr_mont_bytes(coeffs), D, ctx)
    assert I.fr_from_mont(ys) == o.forward_ntt(coeffs + [0] * 5, o.Domain(16))
    assert I.fr_from_mont(polyInverseNTT(ys, D, ctx)) == coeffs + [0] * 5
