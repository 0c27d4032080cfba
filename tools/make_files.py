#!/usr/bin/env python3
"""Writes the benchmark circuit (squaring chain, fake trusted setup on the GPU, seeds of bench.py) as a snarkjs-layout
.zkey / .wtns pair:   python tools/make_files.py <log2n> <dir>   ->  <dir>/c.zkey, <dir>/c.wtns
Inputs of the native tools (tools/g16prove.cpp, tools/ab_prove.cpp)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from nim_groth16_amd import Context
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.files import writeWitness, writeZKey
    from nim_groth16_amd.synthetic import SplitMix64, squaringChain
    log2n, d = int(sys.argv[1]), sys.argv[2]
    os.makedirs(d, exist_ok=True)
    ctx = Context(0)
    t0 = time.time()
    r1cs, wit = squaringChain((1 << log2n) - 2, seed=4)
    rng = SplitMix64(5)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
    zpath, wpath = os.path.join(d, "c.zkey"), os.path.join(d, "c.wtns")
    writeZKey(zpath, zk)
    writeWitness(wpath, wit)
    print(f"wrote {zpath} ({os.path.getsize(zpath) / 2**20:.0f} MiB) and {wpath} in {time.time() - t0:.1f}s", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
