#!/usr/bin/env python3
"""End-to-end from files at 2^20: writes a synthetic .zkey/.wtns pair (fake setup on the GPU), runs the native CLI
(tools/g16prove.cpp, built here) on them with -t, and measures the PCIe-inclusive proof rate (witness handed over
as a host buffer on every call) next to the HBM-resident rate."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nim_groth16_amd import Context, loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.files import writeWitness, writeZKey
    from nim_groth16_amd.synthetic import SplitMix64, squaringChain
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = Context(0)
    r1cs, wit = squaringChain((1 << log2n) - 2, seed=4)
    rng = SplitMix64(5)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
    d = tempfile.mkdtemp(prefix="g16e2e_")
    zpath, wpath = os.path.join(d, "c.zkey"), os.path.join(d, "c.wtns")
    t0 = time.time()
    writeZKey(zpath, zk)
    writeWitness(wpath, wit)
    print(f"wrote {os.path.getsize(zpath) / 2**20:.0f} MiB zkey + {os.path.getsize(wpath) / 2**20:.0f} MiB wtns in {time.time() - t0:.1f}s", flush=True)
    # host-buffer (PCIe-inclusive) vs HBM-resident witness, one proof in flight
    pk = loadProvingKey(zk, ctx)
    wb = F.frSeqToMontBytes(wit)
    d_w = torch.frombuffer(bytearray(wb), dtype=torch.uint8).cuda()
    for name, arg, dev in (("host witness (PCIe-inclusive)", wb, False), ("HBM-resident witness", d_w.data_ptr(), True)):
        for _ in range(3):
            pk.prove(arg, mont=True, device=dev)
        t0 = time.perf_counter()
        for _ in range(10):
            pk.prove(arg, mont=True, device=dev)
        dt = (time.perf_counter() - t0) / 10
        print(f"{name}: {dt * 1e3:.2f} ms/proof (one in flight)", flush=True)
    pk.destroy()
    ctx.close()
    csrc = os.path.join(ROOT, "nim_groth16_amd", "csrc")
    exe = os.path.join(d, "g16prove")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "g16prove.cpp"), "-L" + csrc, "-lg16hip", "-Wl,-rpath," + csrc,
                           "-o", exe])
    out = subprocess.run([exe, "-z", zpath, "-w", wpath, "-o", os.path.join(d, "proof.json"), "-i",
                          os.path.join(d, "public.json"), "-y", "-t"], capture_output=True, text=True)
    print("g16prove:", out.stdout.strip().replace("\n", " | "), out.stderr.strip(), flush=True)
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    os.rmdir(d)


if __name__ == "__main__":
    main()
