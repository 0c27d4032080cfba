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
proof: 3 x 32 n / world bytes in, 768 bytes out per rank."""
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


class ShardedProver:
    """partials_fn(witness_bytes) -> 768-byte record (bytes, or a device uint8 tensor);
    combine_fn(gathered, count, r_bytes, s_bytes) -> (pi_a, pi_b, pi_c).  The defaults are the GPU calls
    of a sharded ProvingKey; tests inject CPU stand-ins to exercise the collective logic under gloo.

    quotient = "tasks" (default for snarkjs-flavour keys on the GPU): the three coset pipelines of the quotient run
    on three different ranks (quotientTaskOwner) and every rank receives only its [h_lo, h_hi) slice of each coset
    vector -- three scatters of 32 n / world bytes per destination -- instead of every rank recomputing all six
    NTTs; "replicated": the round-1 behaviour (no exchange besides the partial records)."""

    def __init__(self, zkey, rank: int, world: int, ctx=None, group=None,
                 partials_fn: Optional[Callable] = None, combine_fn: Optional[Callable] = None,
                 quotient: str = "tasks", pkey=None):
        import torch.distributed as dist
        self.dist, self.group, self.rank, self.world, self.zkey = dist, group, rank, world, zkey
        self.pkey = pkey                # an already loaded key of this rank's shard, or None: load it here
        if partials_fn is None and pkey is None:
            from .prover import loadProvingKey
            self.pkey = loadProvingKey(zkey, ctx, shard_index=rank, shard_count=world)
        self.partials_fn, self.combine_fn = partials_fn, combine_fn
        assert quotient in ("tasks", "replicated")
        self.task_quotient = quotient == "tasks" and self.pkey is not None and zkey.header.flavour == 1
        self._bufs = None
        self._mine = None

    # ---- the task-parallel quotient: begin -> three scatters -> end ------------------------------------------
    def _partials_with_task_quotient(self, witness, mont: bool, mine, device: bool = False):
        import torch
        n, world, rank = self.zkey.header.domainSize, self.world, self.rank
        dev = mine.device
        cpu_group = self.group_is_cpu()
        owned = [v for v in range(3) if quotientTaskOwner(v, world) == rank]
        ranges = [shardRange(n, r, world) for r in range(world)]
        nh = ranges[rank][1] - ranges[rank][0]
        slot = 32 * max(1, max(hi - lo for lo, hi in ranges))     # scatter needs equal chunks: pad to the largest
        if self._bufs is None:      # reused across proofs
            self._bufs = (torch.empty(max(1, len(owned)) * n * 32, dtype=torch.uint8, device=dev),
                          [torch.empty(slot, dtype=torch.uint8, device=dev) for _ in range(3)])
        task_out, slices = self._bufs
        self.pkey.prove_partials_begin(witness, sum(1 << v for v in owned), task_out.data_ptr() if owned else None,
                                       mont=mont, device=device)
        if world == 1:
            ptrs = [task_out.data_ptr() + 32 * n * v for v in range(3)]
        else:
            vec = task_out.cpu() if (cpu_group and owned) else task_out
            for v in range(3):      # the exchange: owner(v) hands every rank its slice of coset vector v
                owner = quotientTaskOwner(v, world)
                # dist.scatter takes a GLOBAL rank as src; owner is a rank of `self.group`
                src = self.dist.get_global_rank(self.group, owner) if self.group is not None else owner
                chunks = None
                if rank == owner:
                    base = 32 * n * owned.index(v)
                    chunks = []
                    for lo, hi in ranges:
                        c = vec[base + 32 * lo: base + 32 * hi]
                        chunks.append(torch.cat([c, c.new_zeros(slot - c.numel())]) if c.numel() != slot else c)
                recv = torch.empty(slot, dtype=torch.uint8) if cpu_group else slices[v]
                self.dist.scatter(recv, chunks, src=src, group=self.group)
                if cpu_group:
                    slices[v].copy_(recv)
            torch.cuda.current_stream(dev).synchronize()      # slices are in HBM before the library's stream reads them
            ptrs = [s_.data_ptr() if nh else None for s_ in slices]
        self.pkey.prove_partials_end(ptrs[0], ptrs[1], ptrs[2], out=mine.data_ptr())

    def prove_raw(self, witness, mont: bool, r: Optional[bytes], s: Optional[bytes], device: bool = False):
        """witness: nvars Fr as bytes, or an int address (pinned host memory, or HBM with device=True);
        r, s: Montgomery mask bytes or None.  -> (pi_a, pi_b, pi_c), identical on every rank."""
        import torch
        on_gpu = self.pkey is not None
        if on_gpu:
            if self._mine is None:
                self._mine = torch.empty(PARTIALS_BYTES, dtype=torch.uint8, device=f"cuda:{self.pkey.ctx.device}")
            mine = self._mine
            if self.task_quotient:
                self._partials_with_task_quotient(witness, mont, mine, device)
            else:
                self.pkey.prove_partials(witness, mont=mont, device=device, out=mine.data_ptr())
            if self.group_is_cpu():
                mine = mine.cpu()               # gloo rehearsal on a one-GPU box: the exchange runs on the host
        else:
            mine = torch.frombuffer(bytearray(self.partials_fn(witness)), dtype=torch.uint8)
        gathered = torch.empty(self.world * PARTIALS_BYTES, dtype=torch.uint8, device=mine.device)
        if self.world > 1:
            self.dist.all_gather_into_tensor(gathered, mine, group=self.group)   # 768 bytes per rank per proof
        else:
            gathered.copy_(mine)
        if on_gpu and gathered.is_cuda:
            torch.cuda.current_stream(gathered.device).synchronize()
            return self.pkey.prove_combine(gathered.data_ptr(), self.world, r, s, device=True)
        if on_gpu:
            return self.pkey.prove_combine(bytes(gathered.numpy()), self.world, r, s)
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

    def group_is_cpu(self) -> bool:
        """True when the process group cannot move CUDA tensors (backend gloo)"""
        if self.world <= 1 or not self.dist.is_initialized():
            return False
        return "nccl" not in str(self.dist.get_backend(self.group)).lower()
