#!/usr/bin/env python3
"""How long the tail of a proof takes once the five MSM partials exist: g16_prove_combine = one small kernel (sum of the
per-rank partials + 5 inversions), a 384-byte copy, and the O(1) mask algebra on the host (prover.nim:279-302: two
254-bit scalar multiplications of GPU-dependent points + a dozen additions, each ending in an inversion)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from nim_groth16_amd import Context, loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import SplitMix64, squaringChain
    ctx = Context(0)
    r1cs, wit = squaringChain((1 << 12) - 2, seed=4)
    rng = SplitMix64(5)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
    pk = loadProvingKey(zk, ctx)
    rec = pk.prove_partials(F.frSeqToMontBytes(wit))
    rb, sb = F.frToMontBytes(rng.fr()), F.frToMontBytes(rng.fr())
    for name, (r, s) in (("random mask", (rb, sb)), ("trivial mask", (None, None))):
        for _ in range(20):
            pk.prove_combine(rec, 1, r, s)
        t0 = time.perf_counter()
        for _ in range(200):
            pk.prove_combine(rec, 1, r, s)
        print(f"g16_prove_combine, {name}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per call", flush=True)


if __name__ == "__main__":
    main()
