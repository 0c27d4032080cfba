import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from nim_groth16_amd import Context, loadProvingKey
from nim_groth16_amd import bn128 as F
from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
from nim_groth16_amd.synthetic import SplitMix64, squaringChain
log2n = 20
ctx = Context(0)
m = (1 << log2n) - 2
r1cs, wit = squaringChain(m, seed=4)
rng = SplitMix64(5)
zkey = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
wb = F.frSeqToMontBytes(wit)
d_w = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda()
for G in (1, 2, 4, 8):
    pk = loadProvingKey(zkey, ctx, shard_index=G - 1, shard_count=G)
    out = torch.empty(768, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        pk.prove_partials(d_w.data_ptr(), mont=True, device=True, out=out.data_ptr())
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        pk.prove_partials(d_w.data_ptr(), mont=True, device=True, out=out.data_ptr())
        ctx.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"shard_count {G}: one rank's prove_partials {dt*1e3:.2f} ms", flush=True)
    pk.destroy()
