"""Object lifetime and sharing rules of the C ABI (include/g16hip.h): registered point sets and keys belong to the
device, not to the context that created them -- any teardown order is legal, and the in-flight proofs of one GPU
share one resident key (ProverPoints are per-circuit constants, reference groth16/zkey_types.nim:36-41)."""
import subprocess
import sys
import threading

import pytest

from oracle import bn254_ref as o
from tests import inputs as I

pytestmark = pytest.mark.gpu

_TEARDOWN = r"""
import ctypes, sys
sys.path.insert(0, {root!r})
from nim_groth16_amd import Context, loadProvingKey
from nim_groth16_amd._lib import load_library
from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
from nim_groth16_amd.synthetic import SplitMix64, squaringChain
lib = load_library()
ctx = Context(0)
r1cs, wit = squaringChain((1 << 7) - 2, seed=4)
rng = SplitMix64(5)
zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
pk = loadProvingKey(zk, ctx)
hs = ctx.register_points(1, zk.pPoints.pointsA1, zk.header.nvars)
from nim_groth16_amd import bn128 as F
proof = pk.prove(F.frSeqToMontBytes(wit))
# the order the C ABI must survive: context first, THEN its key and point set (raw handles, no Python guard)
lib.g16_ctx_destroy(ctx._h); ctx._h = None
lib.g16_pkey_destroy(pk._h); pk._h = None
lib.g16_points_release(hs._h); hs._h = None
# and the key of a dead context's device is still usable from a NEW context of that device
ctx2 = Context(0)
pk2 = loadProvingKey(zk, ctx2)
ctx3 = Context(0)
assert pk2.prove(F.frSeqToMontBytes(wit), ctx=ctx3) == proof
ctx2.close()                      # creating context gone, key alive and in use through ctx3
assert pk2.prove(F.frSeqToMontBytes(wit), ctx=ctx3) == proof
pk2.destroy(); ctx3.close()
print("teardown ok")
"""


def test_teardown_in_any_order_exits_cleanly():
    """g16_ctx_destroy with live g16_pkey / g16_points, then destroying those, must not abort (round-1 core dump at
    interpreter exit: the key's destroy touched its dead context's stream)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _TEARDOWN.format(root=root)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "teardown ok" in r.stdout


def test_one_key_shared_by_concurrent_contexts(ctx, orc):
    """three contexts (private streams + workspaces) prove three DIFFERENT witnesses against ONE resident key from
    three host threads at once; every proof equals the one the creating context produces alone"""
    from nim_groth16_amd import Context, loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import SplitMix64, squaringChain
    m = (1 << 12) - 2
    r1cs, wit0 = squaringChain(m, seed=4)
    wits = [wit0] + [squaringChain(m, seed=4, w0=w0)[1] for w0 in (5, 7)]     # same circuit, other satisfying inputs
    rng = SplitMix64(5)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
    pk = loadProvingKey(zk, ctx)
    r, s = F.frToMontBytes(rng.fr()), F.frToMontBytes(rng.fr())
    wbs = [F.frSeqToMontBytes(w) for w in wits]
    alone = [pk.prove(wb, r=r, s=s) for wb in wbs]
    assert len(set(alone)) == 3
    others = [Context(0) for _ in range(3)]
    got = [[None] * 4 for _ in range(3)]

    def work(i):
        for rep in range(4):
            got[i][rep] = pk.prove(wbs[i], r=r, s=s, ctx=others[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(3):
        assert got[i] == [alone[i]] * 4
    with pytest.raises(ValueError, match="wrong witness length"):
        pk.prove(wbs[0][:-32], ctx=others[0])        # a short host buffer never reaches the C ABI
    for c in others:
        c.close()
    pk.destroy()


def test_sharded_prover_takes_a_parsed_standard_form_witness(ctx, tmp_path):
    """ShardedProver (world 1, real GPU halves) on a parseWitness() witness -- standard form on disk,
    files/witness.nim:14 -- equals generateProofWithMask, incl. the Montgomery publicIO of Proof (prover.nim:238-240)"""
    from nim_groth16_amd import Mask, generateProofWithMask
    from nim_groth16_amd.distributed import ShardedProver
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.files import parseWitness, writeWitness
    rng = o.SplitMix64(61)
    tox = ToxicWaste(*[rng.fr() for _ in range(5)])
    zk = fakeCircuitSetup(R1CS(8, 1, 1, 3, o.toy_r1cs().constraints), tox, 1, ctx)
    wpath = str(tmp_path / "toy.wtns")
    writeWitness(wpath, o.TOY_WITNESS)
    wt = parseWitness(wpath)
    assert wt.std
    mask = Mask(rng.fr(), rng.fr())
    want = generateProofWithMask(0, False, zk, wt, mask, ctx)
    got = ShardedProver(zk, 0, 1, ctx=ctx).prove(wt, mask)
    assert (got.pi_a, got.pi_b, got.pi_c, got.publicIO) == (want.pi_a, want.pi_b, want.pi_c, want.publicIO)
    assert I.fr_from_mont(got.publicIO) == [1, 2023, 1022]


def test_verifier_rejects_non_canonical_encodings(ctx):
    """one proof, one accepted byte encoding: coordinates >= p and public inputs >= r are refused (-5 / -6), not
    silently reduced"""
    from nim_groth16_amd import extractVKey, loadVerifyingKey
    from tests.test_gpu_verifier import _toy
    zk, (good, _) = _toy(ctx)
    dev = loadVerifyingKey(extractVKey(zk), ctx)
    tri = (good.pi_a, good.pi_b, good.pi_c)
    assert dev.verify([tri], good.publicIO) == [1]

    def plus(buf, off, mod):
        v = int.from_bytes(buf[off:off + 32], "little") + mod
        assert v < 1 << 256
        return buf[:off] + v.to_bytes(32, "little") + buf[off + 32:]
    for which, off in ((0, 0), (0, 32), (1, 0), (1, 96), (2, 32)):
        t = list(tri)
        t[which] = plus(t[which], off, o.P)                       # same residue, non-canonical limbs
        assert dev.verify([tuple(t)], good.publicIO) == [-5], (which, off)
    assert dev.verify([tri], plus(good.publicIO, 32, o.R)) == [-6]
    assert dev.verify([tri, tri], good.publicIO + plus(good.publicIO, 64, o.R)) == [1, -6]
    std = I.fr_std_bytes(I.fr_from_mont(good.publicIO))
    assert dev.verify([tri], std, mont=False) == [1]
    assert dev.verify([tri], plus(std, 32, o.R), mont=False) == [-6]


def _small_key(ctx, log2n=10, **shard):
    from nim_groth16_amd import loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import SplitMix64, squaringChain
    r1cs, wit = squaringChain((1 << log2n) - 2, seed=4)
    rng = SplitMix64(5)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
    return zk, loadProvingKey(zk, ctx, **shard), F.frSeqToMontBytes(wit)


def test_pending_begin_is_cancelled_by_any_other_call(ctx):
    """include/g16hip.h: between g16_prove_partials_begin and _end the witness lanes run on the context's sort and
    per-proof buffers; any other compute call first drains them and cancels the pending proof -- round 2 let a second
    call rewrite the offsets the lanes were still reading.  The sequence
    begin -> prove_partials (same context) -> end  must leave a correct record and a clean G16_EINVAL."""
    import torch
    from nim_groth16_amd._lib import G16_EINVAL, G16Error
    zk, pk, wb = _small_key(ctx, 12)
    n = zk.header.domainSize
    try:
        # XYZZ records are not canonical (the order of additions inside a bucket follows the LDS atomics of the sort):
        # records are compared through the proof they combine to
        fin = lambda rec: pk.prove_combine(rec, 1)           # noqa: E731
        want = fin(pk.prove_partials(wb))
        assert want == pk.prove(wb)
        out = torch.empty(3 * n * 32, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            pk.prove_partials_begin(wb, 7, out.data_ptr())
            assert fin(pk.prove_partials(wb)) == want      # cancels the begin, then computes on the drained context
            with pytest.raises(G16Error) as e:
                pk.prove_partials_end(out.data_ptr(), out.data_ptr() + 32 * n, out.data_ptr() + 64 * n)
            assert e.value.code == G16_EINVAL
        # and the regular pair still works afterwards: begin (all three pipelines here) -> end == replicated record
        pk.prove_partials_begin(wb, 7, out.data_ptr())
        assert fin(pk.prove_partials_end(out.data_ptr(), out.data_ptr() + 32 * n, out.data_ptr() + 64 * n)) == want
        # the no-host-sync variant, ordered by the context's stream alone
        rec = torch.empty(768, dtype=torch.uint8, device="cuda")
        pk.prove_partials_begin(wb, 7, out.data_ptr(), nosync=True)
        pk.prove_partials_end(out.data_ptr(), out.data_ptr() + 32 * n, out.data_ptr() + 64 * n, out=rec.data_ptr(),
                              nosync=True)
        ctx.synchronize()
        assert fin(bytes(rec.cpu().numpy())) == want
    finally:
        pk.destroy()


def test_prove_from_a_fresh_host_thread(ctx):
    """every C-ABI call makes the context's device the calling thread's current device (ctx_enter): a new host thread
    starts on device 0, and round 2's g16_prove allocated its result slots before setting the device (ADVICE r02,
    high).  One GPU cannot tell the devices apart, so this covers the thread path; the two-GPU case below runs when
    the box has a second device."""
    zk, pk, wb = _small_key(ctx, 10)
    try:
        want = pk.prove(wb)
        got = []
        th = threading.Thread(target=lambda: got.append(pk.prove(wb)))
        th.start()
        th.join()
        assert got == [want]
    finally:
        pk.destroy()


def test_context_of_second_gpu_from_a_thread_on_device_zero():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from nim_groth16_amd import Context
    c0, c1 = Context(0), Context(1)
    zk, pk0, wb = _small_key(c0, 10)
    _, pk1, _ = _small_key(c1, 10)
    try:
        want = pk0.prove(wb)
        got = []

        def work():                      # this thread's current device is 0; the context lives on device 1
            c = Context(1)
            got.append(pk1.prove(wb, ctx=c))
            c.close()
        th = threading.Thread(target=work)
        th.start()
        th.join()
        assert got == [want]
    finally:
        pk0.destroy()
        pk1.destroy()
        c0.close()
        c1.close()
