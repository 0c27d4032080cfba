"""The device group of the C ABI (g16_group_*: include/g16hip.h): one proof sharded over several "devices" from ONE host
process, one host thread per member inside the library -- the reference's shape, Taskpool tasks over contiguous index
ranges with partial sums added in task order (groth16/bn128/msm.nim:96-122) and three coset-pipeline tasks
(prover.nim:165-173).  On the one-GPU box every member is device 0: the sharding, the slice exchange and the
in-order combination are the real ones, the peer copies are same-device copies.  Proofs must equal g16_prove's on the
unsharded key bit for bit (and through it the C oracle's).  No multi-GPU hardware has run this: no scaling claim."""
import pytest

from oracle import bn254_ref as o

pytestmark = pytest.mark.gpu


def _toxic(seed=5):
    from nim_groth16_amd.fake_setup import ToxicWaste
    from nim_groth16_amd.synthetic import SplitMix64
    rng = SplitMix64(seed)
    return ToxicWaste(*[rng.fr() for _ in range(5)]), rng


@pytest.mark.parametrize("flavour", [1, 0])
def test_group_proofs_equal_the_unsharded_proof(ctx, orc, flavour):
    """G = 1, 2, 3, 8 members; Montgomery and .wtns witnesses; trivial and random masks; both key flavours (JensGroth
    takes the replicated-quotient path: its seventh transform needs the whole vector, prover.nim:142)"""
    from nim_groth16_amd import DeviceGroup, loadGroupKey, loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.fake_setup import fakeCircuitSetup
    from nim_groth16_amd.synthetic import poseidonMerkle
    from tests.parity import check_gpu_proof
    r1cs, wit = poseidonMerkle(12, seed=4)           # rows of 1..25 terms, a third of B1 / B2 at infinity
    tox, rng = _toxic()
    zk = fakeCircuitSetup(r1cs, tox, flavour, ctx)
    wb, ws = F.frSeqToMontBytes(wit), F.frSeqToStdBytes(wit)
    r, s = rng.fr(), rng.fr()
    rb, sb = F.frToMontBytes(r), F.frToMontBytes(s)
    pk = loadProvingKey(zk, ctx)
    want = pk.prove(wb, r=rb, s=sb)
    want0 = pk.prove(wb)
    pk.destroy()
    check_gpu_proof(orc, zk, wit, wb, r, s, want, ctx)
    for G in (1, 2, 3, 8):
        grp = DeviceGroup([0] * G)
        assert grp.size() == G
        gk = loadGroupKey(zk, grp)
        try:
            assert gk.prove(wb, r=rb, s=sb) == want, G
            assert gk.prove(ws, mont=False, r=rb, s=sb) == want, G
            assert gk.prove(wb) == want0, G
            assert gk.prove(wb, r=rb, s=sb) == want, G            # a group is reusable
            with pytest.raises(ValueError):
                gk.prove(wb[:-32])
        finally:
            gk.destroy()
            grp.close()


def test_chained_partial_record_keeps_its_c1_slot_at_infinity(ctx):
    """when C1 and H1 share their bucket set the H accumulation continues C1's bucket sums and the pair is
    reduced once: the record's H1 slot then holds H1 + C1 and its C1 slot XYZZ infinity (all zero) -- g16_prove_combine
    sums both slots, so a non-zero C1 slot there would count C1 twice"""
    from nim_groth16_amd import loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.fake_setup import fakeCircuitSetup
    from nim_groth16_amd.synthetic import squaringChain
    r1cs, wit = squaringChain((1 << 11) - 2, seed=4)
    tox, _ = _toxic()
    zk = fakeCircuitSetup(r1cs, tox, 1, ctx)
    pk = loadProvingKey(zk, ctx)
    try:
        wb = F.frSeqToMontBytes(wit)
        rec = pk.prove_partials(wb)
        assert len(rec) == 768 and rec[640:768] == bytes(128) and rec[512:640] != bytes(128)
        assert pk.prove_combine(rec, 1) == pk.prove(wb)
    finally:
        pk.destroy()


def test_group_key_may_outlive_its_group(ctx):
    from nim_groth16_amd import DeviceGroup, loadGroupKey
    from nim_groth16_amd.fake_setup import fakeCircuitSetup
    from nim_groth16_amd.synthetic import squaringChain
    r1cs, _ = squaringChain((1 << 8) - 2, seed=4)
    tox, _ = _toxic()
    zk = fakeCircuitSetup(r1cs, tox, 1, ctx)
    grp = DeviceGroup([0, 0])
    gk = loadGroupKey(zk, grp)
    grp.close()
    gk.destroy()                                     # must not touch the group


def test_group_argument_errors(ctx):
    from nim_groth16_amd import DeviceGroup
    from nim_groth16_amd._lib import G16Error
    with pytest.raises(G16Error):
        DeviceGroup([0, 9999])                       # no such device
    with pytest.raises(G16Error):
        DeviceGroup([])
    grp = DeviceGroup([0, 0])
    import ctypes
    lib = grp._lib
    out = ctypes.create_string_buffer(256)
    assert lib.g16_group_prove(grp._h, None, b"x", 1, None, None, out) == -1           # G16_EINVAL: no key
    assert b"bad argument" in lib.g16_group_last_error(grp._h)
    grp.close()


def test_ctx_cancel_forgets_a_pending_begin(ctx):
    """g16_ctx_cancel (what ShardedProver and the group call when an exchange failed): drains the lanes, and a later
    _end on that context is refused"""
    import torch
    from nim_groth16_amd import loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd._lib import G16Error
    from nim_groth16_amd.fake_setup import fakeCircuitSetup
    from nim_groth16_amd.synthetic import squaringChain
    r1cs, wit = squaringChain((1 << 10) - 2, seed=4)
    tox, _ = _toxic()
    zk = fakeCircuitSetup(r1cs, tox, 1, ctx)
    pk = loadProvingKey(zk, ctx)
    try:
        wb = F.frSeqToMontBytes(wit)
        out = torch.empty(3 * 32 << 10, dtype=torch.uint8, device="cuda")
        pk.prove_partials_begin(wb, 7, out.data_ptr())
        ctx.cancel()
        with pytest.raises(G16Error):
            pk.prove_partials_end(out.data_ptr(), out.data_ptr() + (32 << 10), out.data_ptr() + (64 << 10))
        ctx.cancel()                                 # idempotent
        assert pk.prove(wb) == pk.prove(wb)
    finally:
        pk.destroy()
