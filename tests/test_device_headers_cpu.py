"""The *device* field/curve headers (nim_groth16_amd/csrc/ff.cuh, ec.cuh) compiled with g++ and checked
against the oracle on the CPU: catches formula errors without a GPU.  (Test build only -- the product never
runs these on the host for the hot path.)"""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import bn254_ref as o

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpu_kernels")


@pytest.fixture(scope="module")
def shim():
    so = os.path.join(HERE, "libffec_shim.so")
    src = os.path.join(HERE, "ffec_shim.cpp")
    import glob
    deps = [src] + glob.glob(os.path.join(HERE, "..", "..", "nim_groth16_amd", "csrc", "*.cuh"))
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
    return ctypes.CDLL(so)


@pytest.fixture(scope="module")
def shim_pair():
    """the same shim built with -DG16_F29_PAIR: the Fp2 products of ec29.cuh go through Fp29::dot_pair (the G2
    accumulate kernel's configuration)"""
    so = os.path.join(HERE, "libffec_shim_pair.so")
    src = os.path.join(HERE, "ffec_shim.cpp")
    import glob
    deps = [src] + glob.glob(os.path.join(HERE, "..", "..", "nim_groth16_amd", "csrc", "*.cuh"))
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-DG16_F29_PAIR", "-shared", "-fPIC", src, "-o", so])
    return ctypes.CDLL(so)


def test_field_formulas(shim):
    rng = random.Random(1)
    for field, mod in ((0, o.P), (1, o.R)):
        Rm = o.MONT % mod
        Ri = pow(Rm, -1, mod)

        def f(op, a, b=0):
            r = ctypes.create_string_buffer(32)
            shim.shim_field_op(field, op, a.to_bytes(32, "little"), b.to_bytes(32, "little"), r)
            return int.from_bytes(r.raw, "little")
        special = [0, 1, mod - 1, mod - 2, 2, (mod + 1) // 2, Rm]
        for t in range(200):
            a = special[t % 7] if t < 49 else rng.randrange(mod)
            b = special[(t // 7) % 7] if t < 49 else rng.randrange(mod)
            assert f(0, a, b) == (a + b) % mod and f(1, a, b) == (a - b) % mod
            assert f(2, a, b) == a * b * Ri % mod and f(3, a) == a * a * Ri % mod
            assert f(4, a) == (-a) % mod and f(5, a) == 2 * a % mod
            assert f(6, a) == a * pow(2, -1, mod) % mod
            assert f(8, a) == a * Ri % mod and f(9, a) == a * Rm % mod
        a = rng.randrange(1, mod)
        assert f(7, a * Rm % mod) == pow(a, -1, mod) * Rm % mod and f(7, 0) == 0
        # the division-step inversion (ff.cuh: 30 steps per batch on signed 30-bit limbs): edge values whose limb
        # patterns stress the sign handling and the early exits, then random ones
        edge = [1, 2, 3, mod - 1, mod - 2, (mod + 1) // 2, (mod - 1) // 2, 1 << 30, (1 << 30) - 1, 1 << 60, (1 << 240) + 1,
                (1 << 253) % mod, Rm, Ri, mod // 3, 0x3fffffff << 30, (1 << 128) - 1, 5 ** 100 % mod]
        for a in edge + [rng.randrange(1, mod) for _ in range(400)]:
            assert f(7, a) == pow(a * Ri % mod, -1, mod) * Rm % mod, hex(a)


def test_fp2_formulas(shim):
    rng = random.Random(2)
    enc = lambda x: o.fp_to_mont_bytes(x[0]) + o.fp_to_mont_bytes(x[1])                         # noqa: E731
    dec = lambda b: (o.fp_from_mont_bytes(b[:32]), o.fp_from_mont_bytes(b[32:64]))              # noqa: E731
    for _ in range(40):
        a = (rng.randrange(o.P), rng.randrange(o.P))
        b = (rng.randrange(o.P), rng.randrange(o.P))
        for op, fn in ((0, o.fp2_add), (1, o.fp2_sub), (2, o.fp2_mul)):
            r = ctypes.create_string_buffer(64)
            shim.shim_fp2_op(op, enc(a), enc(b), r)
            assert dec(r.raw) == fn(a, b)
        r = ctypes.create_string_buffer(64)
        shim.shim_fp2_op(3, enc(a), enc(b), r)
        assert dec(r.raw) == o.fp2_sqr(a)
        shim.shim_fp2_op(7, enc(a), enc(b), r)
        assert dec(r.raw) == o.fp2_inv(a)


@pytest.mark.parametrize("group", [1, 2])
def test_curve_formulas_incl_exceptional_cases(shim, group):
    rng = random.Random(3)
    C, gen = (o.G1, o.GEN1) if group == 1 else (o.G2, o.GEN2)
    enc, dec, psz = (o.g1_to_bytes, o.g1_from_bytes, 64) if group == 1 else (o.g2_to_bytes, o.g2_from_bytes, 128)
    fn = shim.shim_g1_sum if group == 1 else shim.shim_g2_sum

    def gsum(op, pts, n=None):
        r = ctypes.create_string_buffer(psz)
        fn(op, b"".join(enc(p) for p in pts), len(pts) if n is None else n, r)
        return dec(r.raw)
    P1, P2, P3 = (C.mul(rng.randrange(o.R), gen) for _ in range(3))
    cases = [[P1, P2, P3], [P1, P1], [P1, C.neg(P1)], [C.inf, P1, C.inf, P2], [P1, P1, P1, C.neg(P1), P2],
             [C.inf], [], [P1, P2, C.neg(P2), C.neg(P1)], [P1] * 5]
    for pts in cases:
        exp = C.inf
        for q in pts:
            exp = C.add(exp, q)
        assert gsum(0, pts) == exp      # XYZZ += affine (madd), incl. doubling / cancellation / infinity
        assert gsum(1, pts) == exp      # XYZZ += XYZZ
    assert gsum(2, [P1], n=12345) == C.mul(12345, P1) and gsum(2, [P1], n=0) == C.inf


def test_reduced_radix_field(shim):
    """ff29.cuh (9x29-bit limbs, R=2^261) against plain modular arithmetic, through the 8x32 Montgomery layout"""
    rng = random.Random(11)
    R256 = 1 << 256
    edge = [0, 1, o.P - 1, o.P - 2, 2, (1 << 253), (1 << 232) - 1, (1 << 29) - 1, 1 << 29]
    vals = edge + [rng.randrange(o.P) for _ in range(300)]
    r = ctypes.create_string_buffer(32)
    for i, a in enumerate(vals):
        b = vals[(7 * i + 3) % len(vals)]
        am, bm = a * R256 % o.P, b * R256 % o.P
        for op, exp in ((0, a * b), (1, a * a), (2, a), (3, 2 * a * b)):
            shim.shim_f29_op(op, am.to_bytes(32, "little"), bm.to_bytes(32, "little"), r)
            assert int.from_bytes(r.raw, "little") == exp % o.P * R256 % o.P, (op, a, b)


@pytest.mark.parametrize("group", [1, 2])
def test_reduced_radix_accumulate(shim, group):
    """ec29.cuh mixed addition (the accumulate kernels' arithmetic) incl. P+P, P-P, infinity and long lazy chains"""
    rng = random.Random(5)
    C, gen = (o.G1, o.GEN1) if group == 1 else (o.G2, o.GEN2)
    enc, dec, psz = (o.g1_to_bytes, o.g1_from_bytes, 64) if group == 1 else (o.g2_to_bytes, o.g2_from_bytes, 128)
    fn = shim.shim_g1_sum29 if group == 1 else shim.shim_g2_sum29

    def gsum(op, pts):
        r = ctypes.create_string_buffer(psz)
        fn(op, b"".join(enc(p) for p in pts), len(pts), r)
        return dec(r.raw)
    P1, P2, P3 = (C.mul(rng.randrange(o.R), gen) for _ in range(3))
    cases = [[P1, P2, P3], [P1, P1], [P1, C.neg(P1)], [C.inf, P1, C.inf, P2], [P1, P1, P1, C.neg(P1), P2],
             [C.inf], [], [P1, P2, C.neg(P2), C.neg(P1)], [P1] * 5, [P1, P2, C.neg(C.add(P1, P2)), P3],
             [P1, P2, C.add(P1, P2)]]
    pool = [C.mul(rng.randrange(o.R), gen) for _ in range(12)]
    cases.append([rng.choice(pool + [C.inf]) if rng.random() < 0.8 else C.neg(rng.choice(pool)) for _ in range(400)])
    for pts in cases:
        exp = C.inf
        for q in pts:
            exp = C.add(exp, q)
        assert gsum(0, pts) == exp
        assert gsum(1, pts) == exp
        assert gsum(2, pts) == exp      # pairwise tree of general XYZZ + XYZZ additions (bucket reduction)


def test_reduced_radix_accumulate_g2_with_chain_pairs(shim_pair):
    """G2 mixed additions with the interleaved dot-product pairs (Fp29::dot_pair) of the G2 accumulate build"""
    rng = random.Random(6)
    P1, P2, P3 = (o.G2.mul(rng.randrange(o.R), o.GEN2) for _ in range(3))
    pool = [o.G2.mul(rng.randrange(o.R), o.GEN2) for _ in range(6)]
    cases = [[P1, P2, P3], [P1, P1], [P1, o.G2.neg(P1)], [o.G2.inf, P1, o.G2.inf, P2], [P1] * 5,
             [rng.choice(pool + [o.G2.inf]) if rng.random() < 0.8 else o.G2.neg(rng.choice(pool)) for _ in range(150)]]
    for pts in cases:
        exp = o.G2.inf
        for q in pts:
            exp = o.G2.add(exp, q)
        for op in (0, 1):
            r = ctypes.create_string_buffer(128)
            shim_pair.shim_g2_sum29(op, b"".join(o.g2_to_bytes(p) for p in pts), len(pts), r)
            assert o.g2_from_bytes(r.raw) == exp


def test_reduced_radix_interval_model():
    """the worst-case bound proof of the lazy arithmetic (tools/ff29_model.py) still goes through"""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ff29_model", os.path.join(os.path.dirname(HERE), "..", "tools", "ff29_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.main()


def _gt_from_bytes(raw):
    """384-byte flat Fp12 (6 x Fp2 over w^k, Montgomery) -> the oracle's degree-12 polynomial basis"""
    out = [0] * 12
    for k in range(6):
        a = o.fp_from_mont_bytes(raw[64 * k:64 * k + 32])
        b = o.fp_from_mont_bytes(raw[64 * k + 32:64 * k + 64])
        e = o._emb((a, b), k)
        out = [(x + y) % o.P for x, y in zip(out, e)]
    return out


def test_pairing_formulas(shim):
    """pairing.cuh (Miller loop + exact final exponentiation) against the oracle's ate pairing, bit for bit"""
    rng = random.Random(21)
    buf = ctypes.create_string_buffer(384)
    for trial in range(2):
        a, b = rng.randrange(1, o.R), rng.randrange(1, o.R)
        Pt, Q = o.G1.mul(a, o.GEN1), o.G2.mul(b, o.GEN2)
        shim.shim_pairing(0, o.g1_to_bytes(Pt), o.g2_to_bytes(Q), buf)
        f = o.miller_loop(Pt, Q)
        assert _gt_from_bytes(buf.raw) == f
        shim.shim_pairing(1, o.g1_to_bytes(Pt), o.g2_to_bytes(Q), buf)
        assert _gt_from_bytes(buf.raw) == o.final_exp(f)
    # bilinearity through the product path alone: e(aP, bQ) == e(abP, Q)
    shim.shim_pairing(1, o.g1_to_bytes(o.G1.mul(a * b % o.R, o.GEN1)), o.g2_to_bytes(o.GEN2), buf)
    e1 = buf.raw
    shim.shim_pairing(1, o.g1_to_bytes(o.G1.mul(a, o.GEN1)), o.g2_to_bytes(o.G2.mul(b, o.GEN2)), buf)
    assert buf.raw == e1
    # infinity on either side -> 1
    shim.shim_pairing(1, o.g1_to_bytes(o.INF_G1), o.g2_to_bytes(o.GEN2), buf)
    assert _gt_from_bytes(buf.raw) == o._f12_one()


def test_class_bucket_set_of_two_multiplier_tables(shim):
    """msm_class_bucket (msm_params.hpp, shared by the sort kernels): every signed-digit magnitude of a window maps to
    a bucket of the class set whose weight times the table's multiplier is that magnitude; every bucket is used by one
    or two magnitudes; the set is 43 slices of 2^(c-7) buckets."""
    shim.shim_class_buckets_check.restype = ctypes.c_uint32
    for c in (15, 17, 20, 22):
        assert shim.shim_class_buckets_check(c) == 0, c


def test_sort_block_order_is_a_bijection_and_keeps_a_partition_on_one_xcd(shim):
    """bs_block (msm_params.hpp): the workgroup order of bucket_hist / bucket_place -- every (partition, slice) once;
    with a partition count that is a multiple of 8 all eight slices of a partition share the block index modulo 8 (the
    XCD the workgroup is dealt to), so one L2 merges the partition's output lines."""
    shim.shim_bs_block_check.restype = ctypes.c_uint32
    for nparts in (8, 688, 1376, 2752, 1, 3, 10, 43):
        assert shim.shim_bs_block_check(nparts) == 0, nparts


def test_spmv_row_bins(shim):
    """spmv_params.hpp: the nine row bins of the row-balanced buildABC kernel -- monotone in the row length, one trip per
    row everywhere but in the last bin, lane groups at least half full, group sizes that divide a 256-thread workgroup"""
    shim.shim_spmv_bins_check.restype = ctypes.c_uint32
    assert shim.shim_spmv_bins_check(100000) == 0

