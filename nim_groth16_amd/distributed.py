"""One proof sharded over the GPUs of a node: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI).  Every rank keeps a contiguous index range of each ProverPoints array -- the chunks of
msmMultiThreadedG1/G2 (reference groth16/bn128/msm.nim:105-115) with GPUs in place of threads -- computes the
five MSM partials over its range, and the 768-byte partial records are exchanged with ONE all-gather per
proof; every rank then adds the partials in rank order (msm.nim:117-119 `res += sync pending[k]`) and finishes
the proof, so all ranks hold the same, bit-identical proof.  EC addition is not an RCCL reduction operator,
hence all-gather + local add rather than all-reduce.  The quotient is task-parallel: the three coset pipelines
(shiftEvalDomain of Az, Bz, Cz -- three Taskpool tasks in the reference, prover.nim:167-169) run on three different
ranks, and each rank receives just its own [h_lo, h_hi) slice of the three coset vectors (three scatters of
32 n / world bytes per destination), forms its H scalars A1*B1 - C1 and runs its share of the H MSM.  Payloads per
proof: 3 x 32 n / world bytes in, 768 bytes out per rank.

Pipelining (round 3).  A sharded proof has two exchange points (the scatters, the all-gather) and a latency-bound
tail; with one proof in flight the GPUs idle at each of them.  `ShardedProver(depth=2)` keeps `depth` proofs in
flight per rank: every in-flight proof has its own library context (private streams and workspaces; all share the
rank's ONE resident key shard), proof i's witness MSMs and coset pipelines are launched and its scatters enqueued
BEFORE proof i-1 is finished (H MSM, all-gather, combine), and on an RCCL group nothing between `begin` and the
final `combine` blocks the host: the library works on a torch stream (g16_ctx_set_stream), the collectives are
enqueued with async_op=True and ordered against that stream by events (G16_NO_HOST_SYNC).  Every rank issues its
collectives in the same program order -- scatter x3 (i), all-gather (i-1), scatter x3 (i+1), ... -- which is what
RCCL requires.  RCCL has carried this schedule's bytes on ONE rank only (round 5: a world-1 `nccl` group with
`force_collectives`, tests/test_gpu_distributed.py): the API surface -- init with device_id, async scatters with chunk
lists built under a side stream, async all_gather_into_tensor, stream-ordered waits -- is exercised; xGMI is not, and no
scaling curve exists.  The multi-rank tests rehearse the same schedule over gloo, where tensors travel through host
memory and the exchange points do block."""
from __future__ import annotations

from typing import Callable, Optional

from . import bn128 as F
from ._lib import PARTIALS_BYTES
from .prover import Mask, Proof, Witness


def shardRange(N: int, rank: int, world: int):
    """msm.nim:107-115: a = (N*k) div ntasks, b = (N*(k+1)) div ntasks"""
    return (N * rank) // world, (N * (rank + 1)) // world


def quotientTaskOwner(v: int, world: int) -> int:
    """rank that runs coset pipeline v (0: A, 1: B, 2: C) of a sharded proof: the three Taskpool tasks of
    prover.nim:167-169 dealt round-robin over the ranks (world >= 3: one pipeline each on ranks 0, 1, 2)"""
    return v % world


class _Slot:
    """everything one in-flight proof of this rank owns"""

    def __init__(self, ctx, own_ctx: bool):
        self.ctx, self.own_ctx = ctx, own_ctx
        self.stream = None          # torch stream the library context works on (RCCL groups only)
        self.task_out = None        # this rank's coset vectors (owned pipelines x n Fr), HBM
        self.slices = None          # the three received [h_lo, h_hi) slices, HBM
        self.slot_bytes = 0
        self.mine = None            # this rank's 768-byte record, HBM
        self.gathered = None
        self.works = []             # pending async collectives
        self.job = None             # (r, s) of the proof in flight, or None


class ShardedProver:
    """partials_fn(witness_bytes) -> 768-byte record (bytes, or a device uint8 tensor);
    combine_fn(gathered, count, r_bytes, s_bytes) -> (pi_a, pi_b, pi_c).  The defaults are the GPU calls
    of a sharded ProvingKey; tests inject CPU stand-ins to exercise the collective logic under gloo.

    quotient = "tasks" (default for snarkjs-flavour keys on the GPU): the three coset pipelines of the quotient run
    on three different ranks (quotientTaskOwner) and every rank receives only its [h_lo, h_hi) slice of each coset
    vector -- three scatters of 32 n / world bytes per destination -- instead of every rank recomputing all six
    NTTs; "replicated": the round-1 behaviour (no exchange besides the partial records).

    depth: proofs in flight per rank (submit / collect); prove_raw / prove run one proof to completion."""

    def __init__(self, zkey, rank: int, world: int, ctx=None, group=None,
                 partials_fn: Optional[Callable] = None, combine_fn: Optional[Callable] = None,
                 quotient: str = "tasks", pkey=None, depth: int = 2, ctx_factory: Optional[Callable] = None,
                 force_collectives: bool = False):
        import torch.distributed as dist
        self.dist, self.group, self.rank, self.world, self.zkey = dist, group, rank, world, zkey
        self.pkey = pkey                # an already loaded key of this rank's shard, or None: load it here
        if partials_fn is None and pkey is None:
            from .prover import loadProvingKey
            self.pkey = loadProvingKey(zkey, ctx, shard_index=rank, shard_count=world)
        self.partials_fn, self.combine_fn = partials_fn, combine_fn
        assert quotient in ("tasks", "replicated")
        self.task_quotient = quotient == "tasks" and self.pkey is not None and zkey.header.flavour == 1
        self.depth = max(1, depth)
        # force_collectives: a group of ONE rank still issues every collective of the schedule -- three async scatters
        # and one async all-gather per proof, under the slot's stream -- instead of taking the world == 1 shortcuts: the
        # only way to let RCCL carry this code's bytes on a one-GPU box (tests/test_gpu_distributed.py)
        self.force = bool(force_collectives)
        self._ctx_factory = ctx_factory     # extra contexts of the pipeline (tests inject CPU stand-ins)
        self._slots = []
        self._head = 0                  # next slot to submit into
        self._inflight = []             # slots in submission order
        n = zkey.header.domainSize
        self._owned = [v for v in range(3) if quotientTaskOwner(v, world) == rank]
        self._ranges = [shardRange(n, r, world) for r in range(world)]

    # ---- slots --------------------------------------------------------------------------------------------------
    def _slot(self, i: int) -> _Slot:
        import torch
        while len(self._slots) <= i:
            if not self._slots:
                s = _Slot(self.pkey.ctx, False)
            elif self._ctx_factory is not None:
                s = _Slot(self._ctx_factory(), True)
            else:
                from ._lib import Context
                s = _Slot(Context(self.pkey.ctx.device), True)
            if not self.group_is_cpu() and (self.world > 1 or self.force):
                # the library launches on a torch stream, so that collectives enqueued under it are ordered against
                # the library's work by events instead of host waits
                dev = torch.device(f"cuda:{self.pkey.ctx.device}")
                s.stream = torch.cuda.Stream(device=dev)
                s.ctx.set_stream(s.stream.cuda_stream)
            self._slots.append(s)
        return self._slots[i]

    def _buffers(self, s: _Slot):
        import torch
        if s.mine is not None:
            return
        dev = "cpu" if self.pkey.ctx.device is None else f"cuda:{self.pkey.ctx.device}"   # None: CPU stand-in (tests)
        n = self.zkey.header.domainSize
        s.slot_bytes = 32 * max(1, max(hi - lo for lo, hi in self._ranges))   # scatter needs equal chunks: pad to the largest
        s.mine = torch.empty(PARTIALS_BYTES, dtype=torch.uint8, device=dev)
        s.gathered = torch.empty(self.world * PARTIALS_BYTES, dtype=torch.uint8, device=dev)
        if self.task_quotient:
            # only a rank that owns a pipeline holds coset vectors (ranks 3.. of a large group own none)
            s.task_out = torch.empty(max(1, len(self._owned) * n * 32), dtype=torch.uint8, device=dev)
            s.slices = [torch.empty(s.slot_bytes, dtype=torch.uint8, device=dev) for _ in range(3)]

    # ---- stage 1: begin + scatters (returns with the witness lanes running) ----------------------------------------
    def _begin(self, s: _Slot, witness, mont: bool, device: bool):
        import torch
        n, world, rank = self.zkey.header.domainSize, self.world, self.rank
        cpu_group = self.group_is_cpu()
        owned = self._owned
        self.pkey.prove_partials_begin(witness, sum(1 << v for v in owned), s.task_out.data_ptr() if owned else None,
                                       mont=mont, device=device, ctx=s.ctx, nosync=s.stream is not None)
        if world == 1 and not self.force:
            return
        vec = s.task_out.cpu() if (cpu_group and owned) else s.task_out
        with (torch.cuda.stream(s.stream) if s.stream is not None else _null()):
            for v in range(3):      # the exchange: owner(v) hands every rank its slice of coset vector v
                owner = quotientTaskOwner(v, world)
                # dist.scatter takes a GLOBAL rank as src; owner is a rank of `self.group`
                src = self.dist.get_global_rank(self.group, owner) if self.group is not None else owner
                chunks = None
                if rank == owner:
                    base = 32 * n * owned.index(v)
                    chunks = []
                    for lo, hi in self._ranges:
                        c = vec[base + 32 * lo: base + 32 * hi]
                        chunks.append(torch.cat([c, c.new_zeros(s.slot_bytes - c.numel())])
                                      if c.numel() != s.slot_bytes else c)
                if cpu_group:
                    recv = torch.empty(s.slot_bytes, dtype=torch.uint8)
                    self.dist.scatter(recv, chunks, src=src, group=self.group)
                    s.slices[v].copy_(recv)
                else:
                    s.works.append(self.dist.scatter(s.slices[v], chunks, src=src, group=self.group, async_op=True))

    # ---- stage 2: end + all-gather + combine -------------------------------------------------------------------
    def _finish(self, s: _Slot):
        import torch
        n, world, rank = self.zkey.header.domainSize, self.world, self.rank
        r, sm = s.job
        cpu_group = self.group_is_cpu()
        nh = self._ranges[rank][1] - self._ranges[rank][0]
        with (torch.cuda.stream(s.stream) if s.stream is not None else _null()):
            if self.task_quotient:
                if world == 1 and not self.force:
                    ptrs = [s.task_out.data_ptr() + 32 * n * v for v in range(3)]
                else:
                    for w in s.works:
                        w.wait()                 # stream-level wait under RCCL (the host does not block)
                    s.works = []
                    if s.stream is None and s.mine.is_cuda:   # slices are in HBM before the library's own stream reads them
                        torch.cuda.current_stream(s.mine.device).synchronize()
                    ptrs = [t.data_ptr() if nh else None for t in s.slices]
                self.pkey.prove_partials_end(ptrs[0], ptrs[1], ptrs[2], out=s.mine.data_ptr(), ctx=s.ctx,
                                             nosync=s.stream is not None)
            mine = s.mine.cpu() if cpu_group else s.mine
            gathered = torch.empty(world * PARTIALS_BYTES, dtype=torch.uint8) if cpu_group else s.gathered
            if world > 1 or self.force:
                work = self.dist.all_gather_into_tensor(gathered, mine, group=self.group,   # 768 bytes per rank per proof
                                                        async_op=not cpu_group)
                if not cpu_group:
                    work.wait()
            else:
                gathered.copy_(mine)
            if gathered.is_cuda:
                if s.stream is None:
                    torch.cuda.current_stream(gathered.device).synchronize()
                return self.pkey.prove_combine(gathered.data_ptr(), world, r, sm, device=True, ctx=s.ctx)
            return self.pkey.prove_combine(bytes(gathered.numpy()), world, r, sm, ctx=s.ctx)

    # ---- pipelined interface ---------------------------------------------------------------------------------------
    def submit(self, witness, mont: bool, r: Optional[bytes], s: Optional[bytes], device: bool = False):
        """Launch one more sharded proof.  If `depth` proofs are already in flight the oldest one is finished first
        and its proof returned (else None).  Every rank must call submit / collect in the same order."""
        assert self.pkey is not None, "the pipelined interface drives the GPU path"
        done = None
        if len(self._inflight) >= self.depth:
            done = self._finish_oldest()
        slot = self._slot(self._head)
        self._buffers(slot)
        slot.job = (r, s)
        try:
            if self.task_quotient:
                self._begin(slot, witness, mont, device)
            else:
                self.pkey.prove_partials(witness, mont=mont, device=device, out=slot.mine.data_ptr(), ctx=slot.ctx)
        except BaseException:
            # A failed begin / exchange may leave the witness lanes of `begin` running and a proof pending on the
            # context: g16_ctx_cancel drains the main stream AND every lane and forgets the pending proof (a plain
            # synchronize would wait for the main stream only).  The slot stays free: `_head` has not moved.  The
            # original exception is the one the caller sees, whatever the cancel does.
            slot.job, slot.works = None, []
            try:
                slot.ctx.cancel()
            except Exception:           # noqa: BLE001
                pass
            raise
        self._head = (self._head + 1) % self.depth
        self._inflight.append(slot)
        return done

    def _finish_oldest(self):
        slot = self._inflight.pop(0)
        try:
            return self._finish(slot)
        finally:
            slot.job = None

    def collect(self):
        """finish every proof in flight, oldest first -> list of (pi_a, pi_b, pi_c)"""
        out = []
        while self._inflight:
            out.append(self._finish_oldest())
        return out

    def prove_raw(self, witness, mont: bool, r: Optional[bytes], s: Optional[bytes], device: bool = False):
        """witness: nvars Fr as bytes, or an int address (pinned host memory, or HBM with device=True);
        r, s: Montgomery mask bytes or None.  -> (pi_a, pi_b, pi_c), identical on every rank."""
        import torch
        if self.pkey is not None:
            assert not self._inflight, "prove_raw with proofs in flight: use submit / collect"
            self.submit(witness, mont, r, s, device)
            return self.collect()[0]
        mine = torch.frombuffer(bytearray(self.partials_fn(witness)), dtype=torch.uint8)
        gathered = torch.empty(self.world * PARTIALS_BYTES, dtype=torch.uint8, device=mine.device)
        if self.world > 1:
            self.dist.all_gather_into_tensor(gathered, mine, group=self.group)   # 768 bytes per rank per proof
        else:
            gathered.copy_(mine)
        return self.combine_fn(bytes(gathered.numpy()), self.world, r, s)

    def prove(self, wtns: Witness, mask: Mask) -> Proof:
        hdr = self.zkey.header
        assert hdr.nvars * 32 == len(wtns.values), "wrong witness length"      # prover.nim:236
        r = F.frToMontBytes(mask.r) if mask.r % F.primeR else None
        s = F.frToMontBytes(mask.s) if mask.s % F.primeR else None
        # a parsed .wtns is standard form (files/witness.nim:14), a Nim seq[Fr] is Montgomery
        pi_a, pi_b, pi_c = self.prove_raw(wtns.values, not wtns.std, r, s)
        pubIO = wtns.values[: 32 * (hdr.npubs + 1)]
        if wtns.std:                     # Proof.publicIO is seq[Fr]: Montgomery in memory (as prover.py:115-118)
            pubIO = F.frSeqToMontBytes(int.from_bytes(pubIO[i:i + 32], "little") for i in range(0, len(pubIO), 32))
        return Proof(pubIO, pi_a, pi_b, pi_c)

    def close(self):
        """release the extra contexts of the pipeline (the first slot uses the key's own context)"""
        self.collect()
        for s in self._slots:
            if s.own_ctx:
                s.ctx.close()
            elif s.stream is not None:
                s.ctx.set_stream(0)          # back to a stream of its own
        self._slots = []

    def group_is_cpu(self) -> bool:
        """True when the process group cannot move CUDA tensors (backend gloo)"""
        if (self.world <= 1 and not self.force) or not self.dist.is_initialized():
            return False
        return "nccl" not in str(self.dist.get_backend(self.group)).lower()


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
