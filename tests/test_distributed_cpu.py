"""world_size-2 gloo test of the sharded-proof collective logic (no GPU): shard ranges, the single all-gather,
rank-order summation and the mask algebra must reproduce the oracle's unsharded proof bit for bit.  The
per-rank MSM partials and the final combine are oracle stand-ins here (the GPU versions are covered by
tests/test_gpu_prover.py::test_sharded_proof_single_gpu)."""
import os
import socket

import pytest
import torch.multiprocessing as mp

from oracle import bn254_ref as o
from tests import inputs as I

ONE = o.fp_to_mont_bytes(1)


def _xyzz_g1(p):
    return bytes(128) if o.G1.is_inf(p) else o.g1_to_bytes(p) + ONE + ONE


def _xyzz_g2(p):
    return bytes(256) if o.G2.is_inf(p) else o.g2_to_bytes(p) + (ONE + bytes(32)) * 2


def _aff_g1(b):      # XYZZ (x, y, zz, zzz) -> affine; zz == 0 -> infinity
    x, y, zz, zzz = (o.fp_from_mont_bytes(b[i:i + 32]) for i in (0, 32, 64, 96))
    return o.INF_G1 if zz == 0 else (x * pow(zz, -1, o.P) % o.P, y * pow(zzz, -1, o.P) % o.P)


def _aff_g2(b):
    e = lambda k: (o.fp_from_mont_bytes(b[k:k + 32]), o.fp_from_mont_bytes(b[k + 32:k + 64]))   # noqa: E731
    x, y, zz, zzz = e(0), e(64), e(128), e(192)
    return o.INF_G2 if o.fp2_is_zero(zz) else (o.fp2_mul(x, o.fp2_inv(zz)), o.fp2_mul(y, o.fp2_inv(zzz)))


def _setup():
    rng = o.SplitMix64(77)
    tox = o.ToxicWaste(*[rng.fr() for _ in range(5)])
    return o.fake_circuit_setup(o.toy_r1cs(), tox, o.SNARKJS), rng.fr(), rng.fr()


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nim_groth16_amd.distributed import ShardedProver, shardRange
    from nim_groth16_amd.prover import Mask, Witness
    from nim_groth16_amd.zkey_types import GrothHeader, ZKey
    oz, r, s = _setup()
    zk = ZKey(header=GrothHeader("bn128", 1, oz.nvars, oz.npubs, oz.domainSize, oz.logDomainSize))
    wit = o.TOY_WITNESS

    def partials(wbytes):
        w = I.fr_from_mont(wbytes)
        Az, Bz, Cz = o.build_abc(oz.coeffs, oz.domainSize, w)
        qs = o.compute_snarkjs_scalar_coeffs(Az, Bz, Cz)
        zs = w[oz.npubs + 1:]
        wa, wb = shardRange(oz.nvars, rank, world)
        ca, cb = shardRange(oz.nvars - oz.npubs - 1, rank, world)
        ha, hb = shardRange(oz.domainSize, rank, world)
        return (_xyzz_g1(o.G1.msm_naive(w[wa:wb], oz.pointsA1[wa:wb])) +
                _xyzz_g1(o.G1.msm_naive(w[wa:wb], oz.pointsB1[wa:wb])) +
                _xyzz_g2(o.G2.msm_naive(w[wa:wb], oz.pointsB2[wa:wb])) +
                _xyzz_g1(o.G1.msm_naive(qs[ha:hb], oz.pointsH1[ha:hb])) +
                _xyzz_g1(o.G1.msm_naive(zs[ca:cb], oz.pointsC1[ca:cb])))

    def combine(gathered, count, rb, sb):
        rr = o.fr_from_mont_bytes(rb) if rb else 0
        ss = o.fr_from_mont_bytes(sb) if sb else 0
        tot = [o.INF_G1, o.INF_G1, o.INF_G2, o.INF_G1, o.INF_G1]
        for k in range(count):
            rec = gathered[768 * k:768 * (k + 1)]
            tot[0] = o.G1.add(tot[0], _aff_g1(rec[0:128]))
            tot[1] = o.G1.add(tot[1], _aff_g1(rec[128:256]))
            tot[2] = o.G2.add(tot[2], _aff_g2(rec[256:512]))
            tot[3] = o.G1.add(tot[3], _aff_g1(rec[512:640]))
            tot[4] = o.G1.add(tot[4], _aff_g1(rec[640:768]))
        it = iter(tot)
        pr = o.generate_proof_with_mask(oz, wit, rr, ss, msm_g1=lambda c, p: next(it), msm_g2=lambda c, p: next(it),
                                        quotient=lambda *a: [0] * oz.domainSize)
        # generate_proof_with_mask calls msm in the order A1, B1, B2, H1, C1 (prover.nim:282-302) == record order
        return o.g1_to_bytes(pr.pi_a), o.g2_to_bytes(pr.pi_b), o.g1_to_bytes(pr.pi_c)

    sp = ShardedProver(zk, rank, world, partials_fn=partials, combine_fn=combine)
    pr = sp.prove(Witness("bn128", oz.nvars, I.fr_mont_bytes(wit)), Mask(r, s))
    q.put((rank, pr.pi_a, pr.pi_b, pr.pi_c, pr.publicIO))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_rule():
    from nim_groth16_amd.distributed import shardRange
    for N in (0, 1, 7, 8, 1000, (1 << 20) - 3):
        for world in (1, 2, 3, 8):
            edges = [shardRange(N, k, world) for k in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == N
            assert all(edges[k][1] == edges[k + 1][0] for k in range(world - 1))     # contiguous, no overlap
            assert all(b == (N * (k + 1)) // world for k, (_, b) in enumerate(edges))  # msm.nim:109


@pytest.mark.timeout(300)
def test_sharded_proof_world2_gloo():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(rk, 2, port, q)) for rk in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oz, r, s = _setup()
    ref = o.generate_proof_with_mask(oz, o.TOY_WITNESS, r, s)
    for (_, pa, pb, pc, pub) in outs:
        assert (o.g1_from_bytes(pa), o.g2_from_bytes(pb), o.g1_from_bytes(pc)) == (ref.pi_a, ref.pi_b, ref.pi_c)
        assert I.fr_from_mont(pub) == ref.publicIO
    assert o.verify_proof(oz, ref)


# ---- the pipeline of ShardedProver (submit / collect, depth 2) with a CPU stand-in for the sharded key -------------
class _FakeCtx:
    device = None                      # "no GPU": ShardedProver keeps its buffers in host memory
    selftests = 0

    def synchronize(self):
        pass

    cancels = 0

    def selftest(self):                # an ordinary context entry: drains every lane, cancels a pending begin
        self.selftests += 1

    def cancel(self):                  # g16_ctx_cancel: drains every lane and forgets a pending begin
        self.cancels += 1

    def set_stream(self, ptr):
        pass

    def close(self):
        pass


class _FakeKey:
    """what ShardedProver calls on a sharded ProvingKey, computed by the oracle on host memory: begin (this rank's
    coset pipelines into task_out), end (H scalars from the received slices, the five partials), combine"""

    def __init__(self, oz, rank, world, log):
        self.oz, self.rank, self.world, self.log = oz, rank, world, log
        self.ctx = _FakeCtx()
        self.pending = {}

    def prove_partials_begin(self, witness, task_mask, task_out=None, mont=True, device=False, ctx=None, nosync=False):
        import ctypes
        oz = self.oz
        w = I.fr_from_mont(bytes(witness))
        self.log.append(("begin", id(ctx)))
        assert id(ctx) not in self.pending, "a context carries ONE pending begin"
        self.pending[id(ctx)] = w
        Az, Bz, Cz = o.build_abc(oz.coeffs, oz.domainSize, w)
        D, eta = o.Domain(oz.domainSize), o.Domain(2 * oz.domainSize).domainGen
        pos = 0
        for v, vec in enumerate((Az, Bz, Cz)):
            if task_mask & (1 << v):
                buf = I.fr_mont_bytes(o.shift_eval_domain(vec, D, eta))
                ctypes.memmove(task_out + pos, buf, len(buf))
                pos += len(buf)

    def prove_partials_end(self, a1, b1, c1, out=None, ctx=None, nosync=False):
        import ctypes
        from nim_groth16_amd.distributed import shardRange
        oz, rank, world = self.oz, self.rank, self.world
        w = self.pending.pop(id(ctx))
        self.log.append(("end", id(ctx)))
        ha, hb = shardRange(oz.domainSize, rank, world)
        sl = [I.fr_from_mont(ctypes.string_at(p, 32 * (hb - ha))) if hb > ha else [] for p in (a1, b1, c1)]
        qs = [(a * b - c) % o.R for a, b, c in zip(*sl)]
        zs = w[oz.npubs + 1:]
        wa, wb = shardRange(oz.nvars, rank, world)
        # pointsC1 is wire-aligned in the library (public wires hold infinity): the same split of the WIRES
        ca, cb = max(wa - oz.npubs - 1, 0), max(wb - oz.npubs - 1, 0)
        rec = (_xyzz_g1(o.G1.msm_naive(w[wa:wb], oz.pointsA1[wa:wb])) +
               _xyzz_g1(o.G1.msm_naive(w[wa:wb], oz.pointsB1[wa:wb])) +
               _xyzz_g2(o.G2.msm_naive(w[wa:wb], oz.pointsB2[wa:wb])) +
               _xyzz_g1(o.G1.msm_naive(qs, oz.pointsH1[ha:hb])) +
               _xyzz_g1(o.G1.msm_naive(zs[ca:cb], oz.pointsC1[ca:cb])))
        ctypes.memmove(out, rec, 768)

    def prove_combine(self, gathered, count, r=None, s=None, device=False, ctx=None):
        oz = self.oz
        rr = o.fr_from_mont_bytes(r) if r else 0
        ss = o.fr_from_mont_bytes(s) if s else 0
        tot = [o.INF_G1, o.INF_G1, o.INF_G2, o.INF_G1, o.INF_G1]
        for k in range(count):
            rec = gathered[768 * k:768 * (k + 1)]
            tot[0] = o.G1.add(tot[0], _aff_g1(rec[0:128]))
            tot[1] = o.G1.add(tot[1], _aff_g1(rec[128:256]))
            tot[2] = o.G2.add(tot[2], _aff_g2(rec[256:512]))
            tot[3] = o.G1.add(tot[3], _aff_g1(rec[512:640]))
            tot[4] = o.G1.add(tot[4], _aff_g1(rec[640:768]))
        it = iter(tot)
        pr = o.generate_proof_with_mask(oz, o.TOY_WITNESS, rr, ss, msm_g1=lambda c, p: next(it),
                                        msm_g2=lambda c, p: next(it), quotient=lambda *a: [0] * oz.domainSize)
        return o.g1_to_bytes(pr.pi_a), o.g2_to_bytes(pr.pi_b), o.g1_to_bytes(pr.pi_c)


def _pipe_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nim_groth16_amd.distributed import ShardedProver
    from nim_groth16_amd.zkey_types import GrothHeader, ZKey
    oz, r, s = _setup()
    zk = ZKey(header=GrothHeader("bn128", 1, oz.nvars, oz.npubs, oz.domainSize, oz.logDomainSize))
    log = []
    key = _FakeKey(oz, rank, world, log)
    sp = ShardedProver(zk, rank, world, pkey=key, depth=2, ctx_factory=_FakeCtx)
    assert sp.task_quotient
    wb = I.fr_mont_bytes(o.TOY_WITNESS)
    rb, sb = o.fr_to_mont_bytes(r), o.fr_to_mont_bytes(s)
    masks = [(rb, sb), (sb, rb), (None, None), (rb, sb), (sb, rb)]
    out = []
    for m in masks:
        done = sp.submit(wb, True, *m)
        if done is not None:
            out.append(done)
    out += sp.collect()
    # the schedule: proof i's begin is issued BEFORE proof i-1's end (two contexts, alternating)
    kinds = [k for k, _ in log]
    assert kinds == ["begin", "begin", "end", "begin", "end", "begin", "end", "begin", "end", "end"], kinds
    assert len({c for _, c in log}) == 2
    single = sp.prove_raw(wb, True, rb, sb)
    sp.close()
    q.put((rank, out, single))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_pipeline_world2_gloo_depth2():
    """ShardedProver.submit / collect with two proofs in flight per rank under a real 2-rank gloo group (CPU stand-in
    for the sharded key): the three scatters of proof i are issued before the all-gather of proof i-1 on BOTH ranks,
    every context carries one pending begin at most, and all five proofs equal the oracle's, rank by rank."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_pipe_worker, args=(rk, 2, port, q)) for rk in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oz, r, s = _setup()
    want = []
    for (mr, ms) in [(r, s), (s, r), (0, 0), (r, s), (s, r)]:
        ref = o.generate_proof_with_mask(oz, o.TOY_WITNESS, mr, ms)
        want.append((o.g1_to_bytes(ref.pi_a), o.g2_to_bytes(ref.pi_b), o.g1_to_bytes(ref.pi_c)))
    for (_, out, single) in outs:
        assert out == want
        assert single == want[0]


def test_failed_begin_leaves_the_pipeline_consistent():
    """a begin that raises must not consume a slot -- `_head` stays, the slot's context is drained and
    its pending proof forgotten (g16_ctx_cancel), the ORIGINAL exception reaches the caller, and the proofs submitted
    before and after come out in order."""
    from nim_groth16_amd.distributed import ShardedProver
    from nim_groth16_amd.zkey_types import GrothHeader, ZKey
    oz, r, s = _setup()
    zk = ZKey(header=GrothHeader("bn128", 1, oz.nvars, oz.npubs, oz.domainSize, oz.logDomainSize))
    log = []
    key = _FakeKey(oz, 0, 1, log)
    real_begin, calls = key.prove_partials_begin, []

    def flaky_begin(*a, **kw):
        calls.append(1)
        if len(calls) == 2:
            raise RuntimeError("injected begin failure")
        return real_begin(*a, **kw)
    key.prove_partials_begin = flaky_begin
    sp = ShardedProver(zk, 0, 1, pkey=key, depth=2, ctx_factory=_FakeCtx)
    wb = I.fr_mont_bytes(o.TOY_WITNESS)
    rb, sb = o.fr_to_mont_bytes(r), o.fr_to_mont_bytes(s)
    assert sp.submit(wb, True, rb, sb) is None
    with pytest.raises(RuntimeError, match="injected begin failure"):
        sp.submit(wb, True, sb, rb)                      # would have gone into slot 1
    assert sp._head == 1 and len(sp._inflight) == 1 and sp._slots[1].ctx.cancels == 1 and sp._slots[1].job is None
    assert sp.submit(wb, True, sb, rb) is None           # the same slot again
    out = sp.collect()
    want = []
    for (mr, ms) in [(r, s), (s, r)]:
        ref = o.generate_proof_with_mask(oz, o.TOY_WITNESS, mr, ms)
        want.append((o.g1_to_bytes(ref.pi_a), o.g2_to_bytes(ref.pi_b), o.g1_to_bytes(ref.pi_c)))
    assert out == want
    sp.close()
