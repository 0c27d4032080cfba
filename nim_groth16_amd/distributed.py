"""One proof sharded over the GPUs of a node: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI).  Every rank keeps a contiguous index range of each ProverPoints array -- the chunks of
msmMultiThreadedG1/G2 (reference groth16/bn128/msm.nim:105-115) with GPUs in place of threads -- computes the
five MSM partials over its range, and the 768-byte partial records are exchanged with ONE all-gather per
proof; every rank then adds the partials in rank order (msm.nim:117-119 `res += sync pending[k]`) and finishes
the proof, so all ranks hold the same, bit-identical proof.  EC addition is not an RCCL reduction operator,
hence all-gather + local add rather than all-reduce.  buildABC and the quotient NTTs are replicated per rank
(~1 ms, no exchange).  Payload: 768 B per rank per proof -- latency-, not bandwidth-bound on xGMI."""
from __future__ import annotations

from typing import Callable, Optional

from . import bn128 as F
from ._lib import PARTIALS_BYTES
from .prover import Mask, Proof, Witness


def shardRange(N: int, rank: int, world: int):
    """msm.nim:107-115: a = (N*k) div ntasks, b = (N*(k+1)) div ntasks"""
    return (N * rank) // world, (N * (rank + 1)) // world


class ShardedProver:
    """partials_fn(witness_bytes) -> 768-byte record (bytes, or a device uint8 tensor);
    combine_fn(gathered, count, r_bytes, s_bytes) -> (pi_a, pi_b, pi_c).  The defaults are the GPU calls
    of a sharded ProvingKey; tests inject CPU stand-ins to exercise the collective logic under gloo."""

    def __init__(self, zkey, rank: int, world: int, ctx=None, group=None,
                 partials_fn: Optional[Callable] = None, combine_fn: Optional[Callable] = None):
        import torch.distributed as dist
        self.dist, self.group, self.rank, self.world, self.zkey = dist, group, rank, world, zkey
        self.pkey = None
        if partials_fn is None:
            from .prover import loadProvingKey
            self.pkey = loadProvingKey(zkey, ctx, shard_index=rank, shard_count=world)
        self.partials_fn, self.combine_fn = partials_fn, combine_fn

    def prove(self, wtns: Witness, mask: Mask) -> Proof:
        import torch
        hdr = self.zkey.header
        assert hdr.nvars * 32 == len(wtns.values), "wrong witness length"      # prover.nim:236
        r = F.frToMontBytes(mask.r) if mask.r % F.primeR else None
        s = F.frToMontBytes(mask.s) if mask.s % F.primeR else None
        on_gpu = self.pkey is not None
        if on_gpu:
            mine = torch.empty(PARTIALS_BYTES, dtype=torch.uint8, device=f"cuda:{self.pkey.ctx.device}")
            # a parsed .wtns is standard form (files/witness.nim:14), a Nim seq[Fr] is Montgomery
            self.pkey.prove_partials(wtns.values, mont=not wtns.std, out=mine.data_ptr())
            if self.group_is_cpu():
                mine = mine.cpu()               # gloo rehearsal on a one-GPU box: the exchange runs on the host
        else:
            mine = torch.frombuffer(bytearray(self.partials_fn(wtns.values)), dtype=torch.uint8)
        gathered = torch.empty(self.world * PARTIALS_BYTES, dtype=torch.uint8, device=mine.device)
        if self.world > 1:
            self.dist.all_gather_into_tensor(gathered, mine, group=self.group)   # the one exchange per proof
        else:
            gathered.copy_(mine)
        if on_gpu and gathered.is_cuda:
            torch.cuda.current_stream(gathered.device).synchronize()
            pi_a, pi_b, pi_c = self.pkey.prove_combine(gathered.data_ptr(), self.world, r, s, device=True)
        elif on_gpu:
            pi_a, pi_b, pi_c = self.pkey.prove_combine(bytes(gathered.numpy()), self.world, r, s)
        else:
            pi_a, pi_b, pi_c = self.combine_fn(bytes(gathered.numpy()), self.world, r, s)
        pubIO = wtns.values[: 32 * (hdr.npubs + 1)]
        if wtns.std:                     # Proof.publicIO is seq[Fr]: Montgomery in memory (as prover.py:115-118)
            pubIO = F.frSeqToMontBytes(int.from_bytes(pubIO[i:i + 32], "little") for i in range(0, len(pubIO), 32))
        return Proof(pubIO, pi_a, pi_b, pi_c)

    def group_is_cpu(self) -> bool:
        """True when the process group cannot move CUDA tensors (backend gloo)"""
        if self.world <= 1 or not self.dist.is_initialized():
            return False
        return "nccl" not in str(self.dist.get_backend(self.group)).lower()
