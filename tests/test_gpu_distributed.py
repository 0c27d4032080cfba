"""The GPU branch of ShardedProver under a real 2-rank process group on the one-GPU box: both ranks compute their
g16_prove_partials / g16_prove_combine on device 0 with sharded keys (msm.nim:105-115 ranges), the 768-byte records
travel through a gloo all-gather, and every rank must end with the proof of the unsharded key, which is in turn
held to the C oracle.  Both quotient modes: "tasks" (rank 0 runs the A and C coset pipelines, rank 1 the B pipeline,
three scatters of the slices) and "replicated".  Domain 2^16: the per-shard window choice and table sizes are the real ones."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
LOG2N = 16


def _inputs():
    from nim_groth16_amd.synthetic import SplitMix64, squaringChain
    m = (1 << LOG2N) - 2
    r1cs, wit = squaringChain(m, seed=4)
    rng = SplitMix64(5)
    tox = [rng.fr() for _ in range(5)]
    return r1cs, wit, tox, (rng.fr(), rng.fr())


def _rank(rank, world, port, q, quotient):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nim_groth16_amd import Context, Mask, Witness
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.distributed import ShardedProver
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    r1cs, wit, tox, (r, s) = _inputs()
    ctx = Context(0)                                   # both ranks share the box's single GPU
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*tox), 1, ctx)
    sp = ShardedProver(zk, rank, world, ctx=ctx, quotient=quotient)   # partials_fn=None: the real sharded key on the GPU
    assert sp.pkey is not None and sp.group_is_cpu() and sp.task_quotient == (quotient == "tasks")
    out = []
    for std in (False, True):
        vals = F.frSeqToStdBytes(wit) if std else F.frSeqToMontBytes(wit)
        pr = sp.prove(Witness("bn128", len(wit), vals, std=std), Mask(r, s))
        out.append((pr.pi_a, pr.pi_b, pr.pi_c, pr.publicIO))
    # the pipeline: five proofs through submit / collect with two in flight (two contexts on this rank's key shard,
    # proof i's begin + scatters issued before proof i-1's end + all-gather); masks alternate so that neighbours differ
    wb = F.frSeqToMontBytes(wit)
    rb, sb = F.frToMontBytes(r), F.frToMontBytes(s)
    masks = [(rb, sb), (sb, rb), (rb, sb), (None, None), (sb, rb)]
    piped = []
    for (mr, ms) in masks:
        done = sp.submit(wb, True, mr, ms)
        if done is not None:
            piped.append(done)
    assert len(piped) == len(masks) - sp.depth and sp.depth == 2
    piped += sp.collect()
    single = {m: sp.prove_raw(wb, True, *m) for m in set(masks)}
    assert piped == [single[m] for m in masks], "pipelined proofs differ from the one-at-a-time proofs"
    assert piped[0] == out[0][:3]
    sp.close()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()
    sp.pkey.destroy()
    ctx.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("quotient", ["tasks", "replicated"])
def test_sharded_prover_gpu_branch_world2_gloo(ctx, orc, quotient):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank, args=(rk, 2, port, q, quotient)) for rk in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=800) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # reference: the unsharded key in this process, itself bit-exact against the C oracle
    from nim_groth16_amd import Mask, Witness, generateProofWithMask, loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from tests.parity import check_gpu_proof
    r1cs, wit, tox, (r, s) = _inputs()
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*tox), 1, ctx)
    pk = loadProvingKey(zk, ctx)
    wb = F.frSeqToMontBytes(wit)
    want = generateProofWithMask(0, False, zk, Witness("bn128", len(wit), wb), Mask(r, s), ctx, pkey=pk)
    check_gpu_proof(orc, zk, wit, wb, r, s, (want.pi_a, want.pi_b, want.pi_c), ctx)
    for rank in (0, 1):
        for got in outs[rank]:                          # Montgomery and standard-form witness
            assert got == (want.pi_a, want.pi_b, want.pi_c, want.publicIO), rank
    pk.destroy()


# ---- RCCL on the one GPU there is -----------------------------------------------------------
def _rank_nccl_world1(port, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from nim_groth16_amd import Context, loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.distributed import ShardedProver
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    r1cs, wit, tox, (r, s) = _inputs()
    ctx = Context(0)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*tox), 1, ctx)
    sp = ShardedProver(zk, 0, 1, ctx=ctx, quotient="tasks", depth=2, force_collectives=True)
    assert sp.force and not sp.group_is_cpu() and sp.task_quotient
    wb = F.frSeqToMontBytes(wit)
    rb, sb = F.frToMontBytes(r), F.frToMontBytes(s)
    masks = [(rb, sb), (sb, rb), (rb, sb), (None, None)]
    # count the collectives RCCL is asked for
    calls = {"scatter": 0, "all_gather": 0}
    real_scatter, real_gather = dist.scatter, dist.all_gather_into_tensor

    def scatter(*a, **k):
        calls["scatter"] += 1
        assert k.get("async_op") is True and a[0].is_cuda
        return real_scatter(*a, **k)

    def gather(*a, **k):
        calls["all_gather"] += 1
        assert k.get("async_op") is True and a[0].is_cuda
        return real_gather(*a, **k)
    dist.scatter, dist.all_gather_into_tensor = scatter, gather
    piped = []
    for (mr, ms) in masks:
        done = sp.submit(wb, True, mr, ms)
        if done is not None:
            piped.append(done)
    piped += sp.collect()
    assert sp._slots[0].stream is not None and sp._slots[1].stream is not None      # the library worked on torch streams
    dist.scatter, dist.all_gather_into_tensor = real_scatter, real_gather
    assert calls == {"scatter": 3 * len(masks), "all_gather": len(masks)}, calls
    sp.close()
    # the same proofs without any collective, and from the unsharded path
    plain = ShardedProver(zk, 0, 1, ctx=ctx, quotient="tasks", depth=2, pkey=sp.pkey)
    ref = [plain.prove_raw(wb, True, *m) for m in masks]
    plain.close()
    whole = loadProvingKey(zk, ctx)
    direct = [whole.prove(wb, mont=True, r=m[0], s=m[1]) for m in masks]
    whole.destroy()
    q.put((piped, ref, direct, str(dist.get_backend())))
    dist.destroy_process_group()
    sp.pkey.destroy()
    ctx.close()


@pytest.mark.timeout(900)
def test_rccl_carries_the_sharded_schedule_on_a_one_rank_group(ctx, orc):
    """A world-1 `nccl` (= RCCL) group with device_id set, ShardedProver(force_collectives=True, depth=2): every proof
    issues its three scatter(async_op=True) -- chunk lists built under the slot's side stream -- and its
    all_gather_into_tensor(async_op=True), ordered against the library's work by stream waits only.  The proofs must
    equal the collective-free ones, the unsharded key's, and the C oracle's.  This exercises RCCL's API surface as this
    code uses it; it is NOT a multi-GPU measurement (no scaling curve exists)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    p = mpc.Process(target=_rank_nccl_world1, args=(port, q))
    p.start()
    piped, ref, direct, backend = q.get(timeout=800)
    p.join(timeout=120)
    assert p.exitcode == 0 and backend == "nccl"
    assert piped == ref == direct
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from tests.parity import check_gpu_proof
    r1cs, wit, tox, (r, s) = _inputs()
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*tox), 1, ctx)
    check_gpu_proof(orc, zk, wit, F.frSeqToMontBytes(wit), r, s, piped[0], ctx)
