#!/usr/bin/env python3
"""Times the batched GPU verifier (g16_verify): toy-circuit proofs replicated into batches of growing size."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import random
    from nim_groth16_amd import (Context, Mask, Witness, extractVKey, generateProofWithMask, loadVerifyingKey)
    from nim_groth16_amd.fake_setup import R1CS, ToxicWaste, fakeCircuitSetup
    R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    rng = random.Random(1)
    fr = lambda x: (x * (1 << 256) % R).to_bytes(32, "little")   # noqa: E731
    ctx = Context(0)
    # the reference's toy circuit (tests/groth16/testProver.nim:17-55)
    cons = [([], [], [(1, R - 1), (2, 1), (7, 1)]), ([(3, 1)], [(4, 1)], [(6, 1)]), ([(5, 1)], [(6, 1)], [(7, 1)])]
    tw = ToxicWaste(*(rng.randrange(1, R) for _ in range(5)))
    zk = fakeCircuitSetup(R1CS(8, 1, 1, 3, cons), tw, 1, ctx)
    wt = Witness("bn128", 8, b"".join(fr(x) for x in [1, 2023, 1022, 7, 11, 13, 77, 1001]))
    prf = generateProofWithMask(0, False, zk, wt, Mask(rng.randrange(R), rng.randrange(R)), ctx)
    dev = loadVerifyingKey(extractVKey(zk), ctx)
    trip = (prf.pi_a, prf.pi_b, prf.pi_c)
    assert dev.verify([trip], prf.publicIO) == [1]
    for n in (1, 64, 1024, 4096, 16384):
        proofs, pub = [trip] * n, prf.publicIO * n
        dev.verify(proofs, pub)
        t0 = time.perf_counter()
        st = dev.verify(proofs, pub)
        dt = time.perf_counter() - t0
        assert st == [1] * n
        print(f"verify batch {n:6d}: {dt * 1e3:9.2f} ms  {n / dt:10.1f} proofs/s", flush=True)


if __name__ == "__main__":
    main()
