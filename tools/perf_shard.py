"""Per-rank cost of ONE sharded 2^20 proof, measured on one GPU by running a single rank's share
(shard_count = G, shard_index = the most loaded rank):
  replicated : g16_prove_partials            -- buildABC + all six NTTs on every rank (round 1)
  tasks      : g16_prove_partials_begin/_end -- the rank's coset pipeline(s) only; the scatter of the slices is NOT
               included (3 x 32 n / G bytes per rank over xGMI; no multi-GPU box here)
  tasks, rank without a pipeline (G > 3)."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch  # noqa: E402
from nim_groth16_amd import Context, loadProvingKey  # noqa: E402
from nim_groth16_amd import bn128 as F  # noqa: E402
from nim_groth16_amd.distributed import quotientTaskOwner, shardRange  # noqa: E402
from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup  # noqa: E402
from nim_groth16_amd.synthetic import SplitMix64, squaringChain  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
CHECK = "--check" in sys.argv      # also prove shard by shard (all G ranks in turn, task-parallel quotient with the
                                   # exchange done by hand) and require the combined proof == the unsharded key's
n = 1 << log2n
ctx = Context(0)
m = n - 2
r1cs, wit = squaringChain(m, seed=4)
rng = SplitMix64(5)
zkey = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
wb = F.frSeqToMontBytes(wit)
d_w = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda()
out = torch.empty(768, dtype=torch.uint8, device="cuda")
# stand-ins for the coset vectors a rank receives: random field elements (top byte < 0x20: below the modulus) --
# an all-zero stand-in would make the H MSM free
task_out = torch.randint(0, 256, (3 * n, 32), dtype=torch.uint8, device="cuda")
task_out[:, 31] &= 0x1F
task_out = task_out.reshape(-1)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


rb, sb = F.frToMontBytes(rng.fr()), F.frToMontBytes(rng.fr())
want = None
COUNTS = [int(x) for x in os.environ.get("PERF_SHARD_COUNTS", "1,2,4,8").split(",")]   # e.g. "8" under a profiler
for G in COUNTS:
    res = {}
    if CHECK:
        vecs, recs = {}, b""
        keys = []
        for rank in range(G):
            pk = loadProvingKey(zkey, ctx, shard_index=rank, shard_count=G)
            owned = [v for v in range(3) if quotientTaskOwner(v, G) == rank]
            tmp = torch.empty(max(1, len(owned)) * n * 32, dtype=torch.uint8, device="cuda")
            pk.prove_partials_begin(d_w.data_ptr(), sum(1 << v for v in owned), tmp.data_ptr() if owned else None, device=True)
            for i, v in enumerate(owned):
                vecs[v] = tmp[32 * n * i: 32 * n * (i + 1)].clone()
            # one context stands for every rank here, so each rank's proof is finished before the next begins;
            # ranks that need a vector computed by a LATER rank (only G = 2: rank 0 needs B from rank 1) are
            # completed in a second sweep
            keys.append(pk)
            if all(v in vecs for v in range(3)):
                lo, hi = shardRange(n, rank, G)
                torch.cuda.synchronize()
                recs += pk.prove_partials_end(*[vecs[v][32 * lo: 32 * hi].contiguous().data_ptr() for v in range(3)])
            else:
                recs += pk.prove_partials(d_w.data_ptr(), mont=True, device=True)   # replicated quotient: same record
        proof = keys[0].prove_combine(recs, G, rb, sb)
        for pk in keys:
            pk.destroy()
        if want is None:
            want = proof
        assert proof == want, f"sharded proof differs at G = {G}"
        print(f"shard_count {G}: the proof combined from {G} shard records equals the unsharded proof", flush=True)
    for rank in sorted({0, G - 1}):
        pk = loadProvingKey(zkey, ctx, shard_index=rank, shard_count=G)
        lo, hi = shardRange(n, rank, G)
        owned = [v for v in range(3) if quotientTaskOwner(v, G) == rank]
        mask = sum(1 << v for v in owned)
        sl = [task_out.data_ptr() + 32 * (n * v + lo) for v in range(3)]     # stand-ins for the received slices

        def tasks():
            pk.prove_partials_begin(d_w.data_ptr(), mask, task_out.data_ptr() if owned else None, device=True)
            pk.prove_partials_end(sl[0], sl[1], sl[2], out=out.data_ptr())
        res[rank] = (timed(lambda: pk.prove_partials(d_w.data_ptr(), mont=True, device=True, out=out.data_ptr())),
                     timed(tasks), len(owned))
        pk.destroy()
    for rank, (rep, tk, nown) in res.items():
        print(f"shard_count {G} rank {rank}: replicated {rep:.2f} ms | tasks {tk:.2f} ms ({nown} coset pipeline(s) on this rank)",
              flush=True)
