"""The code paths that only non-default settings reach, each in its own process (G16_* knobs are read once per process):
small table windows (the partition sort with fewer than 8 low bits: bucket_place without the fused bookkeeping), the
atomic histogram / scatter sort, wide and single-wave reduce2, compaction of (0,0) points forced on and off, the three
NTT tile geometries, the alternative launch orders, the reduce chunk sizes and the one-lane / quad tails.  Every run proves a small circuit with skewed and infinity-laden
inputs and compares proof, MSMs and NTTs with the oracle bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import sys
sys.path.insert(0, {root!r})
from oracle import bn254_ref as o
from tests import inputs as I
from tests.oracle_c import load_oracle
from tests.parity import check_gpu_proof
from nim_groth16_amd import Context, Mask, Witness, generateProofWithMask, loadProvingKey
from nim_groth16_amd import bn128 as F
from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
from nim_groth16_amd.synthetic import SplitMix64, mixedCircuit
orc = load_oracle()
ctx = Context(0)
ctx.selftest()
for kw in (dict(), dict(zero_pct=0, one_pct=0, lin_pct=60)):
    m = (1 << 11) - 2
    r1cs, wit = mixedCircuit(m, seed=4, **kw)
    rng = SplitMix64(5)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
    pk = loadProvingKey(zk, ctx)
    wb = F.frSeqToMontBytes(wit)
    mask = Mask(rng.fr(), rng.fr())
    pr = generateProofWithMask(0, False, zk, Witness("bn128", m + 2, wb), mask, ctx, pkey=pk)
    check_gpu_proof(orc, zk, wit, wb, mask.r, mask.s, (pr.pi_a, pr.pi_b, pr.pi_c), ctx)
    pk.destroy()
from nim_groth16_amd.synthetic import poseidonMerkle
r1cs, wit = poseidonMerkle(10, seed=4)                          # rows of 1..25 terms: groups of 1..8 lanes in buildABC
rng = SplitMix64(5)
zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
pk = loadProvingKey(zk, ctx)
wb = F.frSeqToMontBytes(wit)
mask = Mask(rng.fr(), rng.fr())
pr = generateProofWithMask(0, False, zk, Witness("bn128", len(wit), wb), mask, ctx, pkey=pk)
check_gpu_proof(orc, zk, wit, wb, mask.r, mask.s, (pr.pi_a, pr.pi_b, pr.pi_c), ctx)
# the .wtns layout: with a value dictionary buildABC reads the v R^2 table, without one the sums are converted per row
std = pk.prove(F.frSeqToStdBytes(wit), mont=False, r=F.frToMontBytes(mask.r), s=F.frToMontBytes(mask.s))
assert std == (pr.pi_a, pr.pi_b, pr.pi_c)
assert pk.build_abc(F.frSeqToStdBytes(wit), mont=False) == pk.build_abc(wb)
import os
assert (pk.abc_info()["dict_values"] > 0) == (os.environ.get("G16_ABC_DICT", "1") != "0")
pk.destroy()
for group, n in ((1, 3000), (2, 700), (1, 64), (1, 1)):
    ks = I.uniform_scalars(n, 11)
    sc = I.circom_like_scalars(n, 12)
    psz = 64 * group
    pts = bytearray(orc.fixed_base(group, I.fr_mont_bytes(ks)))
    for i in range(0, n, 3):
        pts[psz * i: psz * (i + 1)] = bytes(psz)               # every third point at infinity
    pts = bytes(pts)
    want = orc.msm(group, I.fr_mont_bytes(sc), pts)
    assert ctx.msm(group, I.fr_mont_bytes(sc), pts, n) == want
    h = ctx.register_points(group, pts, n)
    assert ctx.msm_points(h, I.fr_mont_bytes(sc)) == want
    assert ctx.msm_points(h, I.fr_std_bytes(sc), mont=False) == want
    h.release()
for log2n in (1, 5, 9, 12):
    xb = I.fr_mont_bytes(I.uniform_scalars(1 << log2n, 20 + log2n))
    assert ctx.ntt(xb, log2n, False) == orc.ntt(xb, log2n, inverse=False)
    assert ctx.ntt(xb, log2n, True) == orc.ntt(xb, log2n, inverse=True)
    a, b, c = (I.fr_mont_bytes(I.uniform_scalars(1 << log2n, s)) for s in (31, 32, 33))
    assert ctx.quotient(a, b, c, log2n, 1) == orc.quotient_snarkjs(a, b, c, log2n)
    assert ctx.quotient(a, b, c, log2n, 0) == orc.quotient_jensgroth(a, b, c, log2n)
print("knobs ok")
"""

KNOBS = [
    {"G16_TABLE_WINDOW": "6", "G16_MSM_WINDOW": "5"},          # < 8 low bits: unfused bucket_place + scan kernels
    {"G16_TABLE_WINDOW": "9", "G16_MSM_WINDOW": "9"},          # exactly 8 low bits, one partition
    {"G16_TABLE_WINDOW": "14", "G16_MSM_WINDOW": "12", "G16_R2_WIDTH": "0"},
    {"G16_MSM_SORT": "a"},                                      # global-atomic histogram + scatter
    {"G16_R2_WIDTH": "2", "G16_INF_COMPACT": "0"},             # single-wave reduce2; compaction for every set
    {"G16_INF_COMPACT": "101", "G16_MSM_SEG": "8"},            # compaction off; short segments: many split buckets
    {"G16_NTT_TILE": "1024", "G16_QUOTIENT_FIRST": "1", "G16_LANES_AFTER_QUOTIENT": "1"},
    {"G16_NTT_TILE": "4096", "G16_G1_LANES": "302", "G16_STREAM_PRIO": "nnnnnn"},
    {"G16_G1_BATCH": "1"},                                      # A1, B1, C1 as one batched launch sequence (round 4)
    {"G16_G1_BATCH": "1", "G16_CHAIN_CH": "0", "G16_INF_COMPACT": "0", "G16_MSM_SEG": "8"},   # ... with own sorts, split buckets
    {"G16_CHAIN_CH": "0"},                                      # C1 and H1 as two separate MSMs (rounds 1-3)
    {"G16_MTAB": "1"},                                          # one table per window, plain bucket set (rounds 1-3)
    {"G16_TABLE_WINDOW": "15"},                                 # the smallest window with the class bucket set
    {"G16_TABLE_WINDOW": "17", "G16_MSM_SORT": "a"},           # class bucket set through the global-atomic sort
    {"G16_TABLE_WINDOW": "16", "G16_MSM_SEG": "8", "G16_G1_BATCH": "1"},   # class set + split buckets + batched tails
    {"G16_CHAIN_CH": "1", "G16_MSM_SEG": "8", "G16_INF_COMPACT": "0"},   # the chain across split buckets and own sorts
    # small bucket sets default to 4-bucket reduce chunks, quad-cooperative reduce2 / fold and B2 first: the other side
    {"G16_TAIL_QUAD": "0", "G16_RED_CHUNK": "16", "G16_G2_FIRST": "0"},
    {"G16_TAIL_QUAD": "1", "G16_RED_CHUNK": "8", "G16_G2_FIRST": "2", "G16_TABLE_WINDOW": "16"},
    {"G16_TAIL_QUAD": "1", "G16_RED_CHUNK": "2", "G16_MTAB": "1"},       # quad reduce2 in front of the one-lane merged fold
    {"G16_TAIL_QUAD": "1", "G16_MSM_WINDOW": "16"},                     # 2048 chunks per window: the 128-slot G1 variant
    {"G16_ABC_DICT": "0"},                                      # buildABC on 32-byte values although a dictionary would do
    {"G16_CZ_FLY": "0"},                                        # Cz written by a kernel of its own (round 5, first half)
    {"G16_CZ_FLY": "0", "G16_ABC_DICT": "0", "G16_QUOTIENT_FIRST": "0"},   # ... and the launch order of rounds 1-4
    {"G16_QUOTIENT_FIRST": "0", "G16_G2_FIRST": "1"},
    {"G16_CU_SPLIT": "8"},                                      # main stream on 8 CUs per XCD, lanes on the rest (H on the spare lane)
    {"G16_CU_SPLIT": "4", "G16_HEAVY_GRID": "64", "G16_QUOTIENT_FIRST": "0"},
]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: ",".join(f"{a[4:]}={b}" for a, b in k.items()))
def test_non_default_paths_are_bit_exact(knobs):
    env = {k: v for k, v in os.environ.items() if not k.startswith("G16_")}
    env.update(knobs)
    r = subprocess.run([sys.executable, "-c", _SCRIPT.format(root=ROOT)], env=env, capture_output=True, text=True,
                       timeout=800)
    assert r.returncode == 0 and "knobs ok" in r.stdout, r.stderr[-3000:]
