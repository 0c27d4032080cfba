"""GPU parity: HIP MSM (through the C ABI) == oracle, bit-exact on the canonical affine encoding.
Mirrors what the reference can check about msmMultiThreadedG1/G2 (groth16/bn128/msm.nim:89-158)."""
import pytest

from oracle import bn254_ref as o
from tests import inputs as I

pytestmark = pytest.mark.gpu

PSZ = {1: 64, 2: 128}
GEN = {1: o.g1_to_bytes(o.GEN1), 2: o.g2_to_bytes(o.GEN2)}
INF = {1: bytes(64), 2: bytes(128)}


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("n", [1, 2, 63, 127, 128, 1000])
def test_msm_small_vs_naive_definition(ctx, orc, group, n):
    ks, pts = I.points_with_logs(orc, group, n, seed=100 + n)
    sc = I.uniform_scalars(n, seed=200 + n)
    sb = I.fr_mont_bytes(sc)
    got = ctx.msm(group, sb, pts, n)
    assert got == orc.msm_naive(group, sb, pts)          # msm.nim:162-198 definition
    assert got == I.expected_from_logs(group, sc, ks)    # closed form


@pytest.mark.parametrize("group", [1, 2])
def test_msm_empty_is_infinity(ctx, group):
    assert ctx.msm(group, b"", b"", 0) == INF[group]      # msm.nim:117 res = infG1


@pytest.mark.parametrize("group", [1, 2])
def test_msm_edge_scalars_and_points(ctx, orc, group):
    psz = PSZ[group]
    ks, pts = I.points_with_logs(orc, group, 8, seed=7)
    P = [pts[psz * i:psz * (i + 1)] for i in range(8)]
    C = o.G1 if group == 1 else o.G2
    dec = o.g1_from_bytes if group == 1 else o.g2_from_bytes
    enc = o.g1_to_bytes if group == 1 else o.g2_to_bytes
    negP0 = enc(C.neg(dec(P[0])))
    cases = {
        "all zero scalars": ([0] * 8, P),
        "scalar one": ([1] * 8, P),
        "scalar r-1": ([o.R - 1] * 8, P),
        "repeated point": ([5, 5, 5, 9], [P[0], P[0], P[0], P[1]]),
        "P and -P same bucket": ([77, 77, 3], [P[0], negP0, P[2]]),
        "infinity inputs": ([11, 12, 13], [INF[group], P[1], INF[group]]),
        "cancels to infinity": ([9, 9], [P[0], negP0]),
        "half window edge": ([1 << 15, (1 << 15) + 1, (1 << 16) - 1, 1 << 16], P[:4]),
        "doubling inside bucket": ([3, 3], [P[4], P[4]]),
    }
    for name, (sc, pl) in cases.items():
        sb, pb = I.fr_mont_bytes(sc), b"".join(pl)
        assert ctx.msm(group, sb, pb, len(sc)) == orc.msm_naive(group, sb, pb), name


@pytest.mark.parametrize("group", [1, 2])
def test_msm_std_scalars_flag(ctx, orc, group):
    n = 300
    ks, pts = I.points_with_logs(orc, group, n, seed=31)
    sc = I.uniform_scalars(n, seed=32)
    assert ctx.msm(group, I.fr_std_bytes(sc), pts, n, mont=False) == I.expected_from_logs(group, sc, ks)


@pytest.mark.parametrize("group,n", [(1, 1 << 14), (2, 1 << 13)])
@pytest.mark.parametrize("dist", ["uniform", "circom"])
def test_msm_mid_vs_oracle_pippenger(ctx, orc, group, n, dist):
    ks, pts = I.points_with_logs(orc, group, n, seed=41)
    sc = I.uniform_scalars(n, 42) if dist == "uniform" else I.circom_like_scalars(n, 43)
    sb = I.fr_mont_bytes(sc)
    got = ctx.msm(group, sb, pts, n)
    assert got == orc.msm(group, sb, pts)
    assert got == I.expected_from_logs(group, sc, ks)


@pytest.mark.parametrize("group,n", [(1, 20000), (2, 6000)])
def test_msm_skewed_single_bucket(ctx, orc, group, n):
    """every scalar equal: one bucket per window holds all N entries (heavy-bucket path: workgroup LDS tree of
    reduced-radix partials), one-shot and registered; then a mix of a few heavy and many light buckets"""
    ks, pts = I.points_with_logs(orc, group, n, seed=51)
    for sc in ([0x1234567] * n, [(7, 1 << 40, 3)[i % 3] if i % 5 else I.uniform_scalars(1, i)[0] for i in range(n)]):
        exp = I.expected_from_logs(group, sc, ks)
        assert ctx.msm(group, I.fr_mont_bytes(sc), pts, n) == exp
        h = ctx.register_points(group, pts, n)
        try:
            assert ctx.msm_points(h, I.fr_mont_bytes(sc)) == exp
        finally:
            h.release()


def test_msm_partials_sum(ctx, orc):
    """msm.nim:105-119: contiguous chunks, partials added -> same point (multi-GPU sharding path)."""
    import torch
    n = 3000
    ks, pts = I.points_with_logs(orc, 1, n, seed=61)
    sc = I.uniform_scalars(n, 62)
    sb = I.fr_mont_bytes(sc)
    full = ctx.msm(1, sb, pts, n)
    parts = b""
    for a, b in ((0, 1000), (1000, 1001), (1001, 3000)):
        ds = torch.frombuffer(bytearray(sb[32 * a:32 * b]), dtype=torch.uint8).cuda()
        dp = torch.frombuffer(bytearray(pts[64 * a:64 * b]), dtype=torch.uint8).cuda()
        parts += ctx.msm(1, ds.data_ptr(), dp.data_ptr(), b - a, partial=True)
    assert ctx.sum_partials(1, parts, 3) == full


@pytest.mark.parametrize("group,n", [(1, 1), (1, 700), (1, 1 << 13), (2, 700), (2, 1 << 12)])
def test_msm_registered_points_tables(ctx, orc, group, n):
    """registered point set (precomputed 2^(cw) P_i tables) == plain MSM == oracle; also std scalars,
    infinity points inside the set, and reuse of one registration across several scalar vectors."""
    psz = PSZ[group]
    ks, pts = I.points_with_logs(orc, group, n, seed=71)
    if n >= 700:   # plant infinities (snarkjs keys contain (0,0) for unused wires, curves.nim:95-98)
        pts = bytearray(pts)
        for i in (3, 77, n - 1):
            pts[psz * i:psz * (i + 1)] = bytes(psz)
            ks[i] = 0
        pts = bytes(pts)
    h = ctx.register_points(group, pts, n)
    try:
        for seed, dist in ((72, "uniform"), (73, "circom"), (74, "uniform")):
            sc = I.uniform_scalars(n, seed) if dist == "uniform" else I.circom_like_scalars(n, seed)
            got = ctx.msm_points(h, I.fr_mont_bytes(sc))
            assert got == I.expected_from_logs(group, sc, ks), (dist, seed)
            assert got == ctx.msm(group, I.fr_mont_bytes(sc), pts, n)
        sc = I.uniform_scalars(n, 75)
        assert ctx.msm_points(h, I.fr_std_bytes(sc), mont=False) == I.expected_from_logs(group, sc, ks)
        part = ctx.msm_points(h, I.fr_mont_bytes(sc), partial=True)
        assert ctx.sum_partials(group, part, 1) == I.expected_from_logs(group, sc, ks)
    finally:
        h.release()


@pytest.mark.parametrize("group", [1, 2])
def test_msm_registered_repeated_and_opposite_points(ctx, orc, group):
    """snarkjs keys repeat points; with equal scalars they meet in ONE bucket of the merged bucket set, where the
    reduced-radix accumulate must take its exact P+P (doubling) and P-P (infinity) paths (ec29.cuh)."""
    psz = PSZ[group]
    _, pts = I.points_with_logs(orc, group, 4, seed=81)
    P = [pts[psz * i:psz * (i + 1)] for i in range(4)]
    C = o.G1 if group == 1 else o.G2
    dec, enc = (o.g1_from_bytes, o.g1_to_bytes) if group == 1 else (o.g2_from_bytes, o.g2_to_bytes)
    neg0 = enc(C.neg(dec(P[0])))
    pl = [P[0]] * 5 + [neg0] * 2 + [P[1], P[1], INF[group], P[2], neg0, P[0]] + [P[3]] * 3
    rng = o.SplitMix64(82)
    k = rng.fr()
    for sc in ([k] * len(pl), [3] * len(pl), [k] * 7 + [rng.fr() for _ in range(len(pl) - 7)]):
        sb, pb = I.fr_mont_bytes(sc), b"".join(pl)
        h = ctx.register_points(group, pb, len(pl))
        try:
            assert ctx.msm_points(h, sb) == orc.msm_naive(group, sb, pb)
        finally:
            h.release()


@pytest.mark.parametrize("group", [1, 2])
def test_msm_equal_and_opposite_bucket_sums(ctx, orc, group):
    """the SAME point alone in neighbouring buckets (small distinct scalars): the running sums of msm_reduce1 then add
    equal / opposite bucket sums -- the exact P+P and P-P paths of the general reduced-radix addition (Ec29::add)"""
    psz = PSZ[group]
    _, pts = I.points_with_logs(orc, group, 2, seed=91)
    P0, P1 = pts[:psz], pts[psz:]
    C = o.G1 if group == 1 else o.G2
    dec, enc = (o.g1_from_bytes, o.g1_to_bytes) if group == 1 else (o.g2_from_bytes, o.g2_to_bytes)
    neg0 = enc(C.neg(dec(P0)))
    cases = [([5, 6], [P0, P0]), ([6, 5], [P0, neg0]), (list(range(1, 17)), [P0] * 16),
             (list(range(1, 33)), [P0, neg0] * 16), ([3, 3, 4, 4, 7], [P0, P1, P0, P1, neg0]),
             ([1, 2, 4, 8, 16, 32], [P0] * 6)]
    for sc, pl in cases:
        sb, pb = I.fr_mont_bytes(sc), b"".join(pl)
        exp = orc.msm_naive(group, sb, pb)
        assert ctx.msm(group, sb, pb, len(sc)) == exp, sc
        h = ctx.register_points(group, pb, len(pl))
        try:
            assert ctx.msm_points(h, sb) == exp, sc
        finally:
            h.release()


def test_msm_registered_empty(ctx):
    h = ctx.register_points(1, b"", 0)
    assert ctx.msm_points(h, b"") == INF[1]
    h.release()


def test_msm_atomic_sort_path_still_agrees(ctx, orc, monkeypatch):
    """the global-atomic histogram/scatter path (used when the partition sort does not apply) stays correct"""
    n = 5000
    ks, pts = I.points_with_logs(orc, 1, n, seed=81)
    sc = I.circom_like_scalars(n, 82)
    want = I.expected_from_logs(1, sc, ks)
    assert ctx.msm(1, I.fr_mont_bytes(sc), pts, n) == want
    monkeypatch.setenv("G16_MSM_SORT", "atomic")
    assert ctx.msm(1, I.fr_mont_bytes(sc), pts, n) == want
