#!/usr/bin/env python3
"""BASELINE config 5's workload SHAPE on one GPU: a Poseidon-shaped Merkle-inclusion circuit (nim_groth16_amd/synthetic.py:
rows of 1..25 terms, ncoeffs ~ 12 n, a third of the wires absent from B) next to the squaring chain of config 3, same box,
same session: shape of the key, stand-alone time of buildABC (`abc_spmv`, HIP events inside the library), proofs/s with
three proofs in flight, single-proof latency.  The proof of the first witness is checked against the C oracle bit for bit.

  python tools/perf_poseidon.py [--log2n 20] [--steps 96] [--only abc]     (--only abc: the rocprofv3 --pmc workload)"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
from nim_groth16_amd import Context, Mask, loadProvingKey  # noqa: E402
from nim_groth16_amd import bn128 as F  # noqa: E402
from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup  # noqa: E402
from nim_groth16_amd.synthetic import SplitMix64, poseidonMerkle, squaringChain  # noqa: E402


def shape(r1cs_or_coeffs, zk, pk):
    hdr = zk.header
    info, inf = pk.abc_info(), pk.inf_counts()
    n = hdr.domainSize
    print(f"   nvars {hdr.nvars}, domain 2^{hdr.logDomainSize}, ncoeffs {info['ncoeffs']} = {info['ncoeffs']/n:.2f} n; "
          f"value dictionary: {info['dict_values'] or 'none'}")
    print("   rows of A and B by number of terms L: " +
          ", ".join(f"{g}: {c}" for g, c in info["rows_by_terms"].items() if c))
    print(f"   points at infinity: A1 {inf['A1']}, B1 {inf['B1']} ({100*inf['B1']/hdr.nvars:.1f} %), B2 {inf['B2']}, "
          f"C1 {inf['C1']}, H1 {inf['H1']}; compacted entry lists: A {inf['compact_A']}, B {inf['compact_B']}")


def abc_time(ctx, pk, wb, reps=5, mont=True):
    pk.build_abc(wb, mont=mont)
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(reps):
        pk.build_abc(wb, mont=mont)
    rep = ctx.profile_report()
    ctx.profile(False)
    return rep["abc_spmv"]["total_ms"] / rep["abc_spmv"]["calls"], rep["abc_cz"]["total_ms"] / rep["abc_cz"]["calls"]


def throughput(ctxs, pk, h_w, rb, sb, steps, repeats=3):
    inflight = len(ctxs)
    errs = []

    def run(count):
        def work(j):
            try:
                torch.cuda.set_device(0)
                for i in range(j, count, inflight):
                    pk.prove(h_w[i % len(h_w)].data_ptr(), mont=False, r=rb, s=sb, ctx=ctxs[j])
            except BaseException as e:      # noqa: BLE001
                errs.append(repr(e))
        th = [threading.Thread(target=work, args=(j,)) for j in range(inflight)]
        [t.start() for t in th]
        [t.join() for t in th]
        if errs:
            raise SystemExit("FAIL: " + "; ".join(errs))
    run(12)
    vals = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        vals.append(steps / (time.perf_counter() - t0))
    t0 = time.perf_counter()
    for i in range(3):
        pk.prove(h_w[i % len(h_w)].data_ptr(), mont=False, r=rb, s=sb, ctx=ctxs[0])
    lat = (time.perf_counter() - t0) / 3 * 1e3
    return sorted(vals)[len(vals) // 2], vals, lat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--only", choices=["all", "abc", "poseidon"], default="all")
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    ctx = Context(0)
    ctx.selftest()
    ctxs = [ctx, Context(0), Context(0)]
    rng = SplitMix64(5)
    tox = ToxicWaste(*[rng.fr() for _ in range(5)])
    mask = Mask(rng.fr(), rng.fr())
    rb, sb = F.frToMontBytes(mask.r), F.frToMontBytes(mask.s)
    n = 1 << args.log2n
    circuits = [("poseidon-merkle (config 5's shape)", lambda: poseidonMerkle(args.log2n, seed=4))]
    if args.only == "all":
        circuits.append(("squaring chain (config 3)", lambda: squaringChain(n - 2, seed=4)))
    for name, gen in circuits:
        t0 = time.time()
        r1cs, wit = gen()
        t1 = time.time()
        zk = fakeCircuitSetup(r1cs, tox, 1, ctx)
        t2 = time.time()
        pk = loadProvingKey(zk, ctx)
        print(f"== {name}, domain 2^{args.log2n}: circuit + witness {t1-t0:.1f}s, fake setup {t2-t1:.1f}s, key upload + "
              f"tables {time.time()-t2:.1f}s", flush=True)
        shape(r1cs, zk, pk)
        wb = F.frSeqToMontBytes(wit)
        ws = F.frSeqToStdBytes(wit)
        ms, cz = abc_time(ctx, pk, wb)
        ms_std, cz_std = abc_time(ctx, pk, ws, mont=False)
        info = pk.abc_info()
        per = 8 if info["dict_values"] else 36
        alg = info["ncoeffs"] * per + 2 * 32 * n + 16 * n + 32 * zk.header.nvars
        print(f"   abc_cz (Cz = Az * Bz; .wtns layout: + Az, Bz to Montgomery form) stand-alone: {cz:.4f} / {cz_std:.4f} ms")
        print(f"   abc_spmv stand-alone: {ms:.4f} ms (Montgomery witness), {ms_std:.4f} ms (.wtns layout); compulsory "
              f"bytes {alg/1e6:.1f} MB ({per} B/entry + 64 B/row out + 16 B/row offsets and row ids + the witness once) -> "
              f"{alg/ms/1e6:.0f} GB/s; {info['ncoeffs']/ms/1e6:.2f} G entries/s", flush=True)
        if args.only == "abc":
            continue
        h_w = []
        for _ in range(1):
            t = torch.empty(32 * len(wit), dtype=torch.uint8).pin_memory()
            t.copy_(torch.frombuffer(bytearray(ws), dtype=torch.uint8))
            h_w.append(t)
        med, vals, lat = throughput(ctxs, pk, h_w, rb, sb, args.steps)
        print(f"   proofs/s, 3 in flight, host witness: {med:.2f} (runs {', '.join(f'{v:.2f}' for v in vals)}); "
              f"single proof {lat:.2f} ms", flush=True)
        ctx.profile(True)
        ctx.profile_reset()
        proof = pk.prove(h_w[0].data_ptr(), mont=False, r=rb, s=sb)
        rep = ctx.profile_report()
        ctx.profile(False)
        tot = sum(v["total_ms"] for v in rep.values())
        print(f"   one proof alone, kernel time by HIP events ({tot:.2f} ms summed over streams):")
        for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])[:14]:
            print(f"      {k:24s} {v['total_ms']:8.3f} ms  ({v['calls']} launches)")
        from nim_groth16_amd import Proof, extractVKey, verifyProof
        pio = wb[:32 * (zk.header.npubs + 1)]
        ok = verifyProof(extractVKey(zk), Proof(pio, *proof), ctx)
        print(f"   GPU verifier (pairing equation, verifier.nim:31-52): {'accepts' if ok else 'REJECTS'} the proof", flush=True)
        assert ok
        if not args.no_oracle:
            from tests.oracle_c import load_oracle
            from tests.parity import check_gpu_proof
            cpu_s = check_gpu_proof(load_oracle(), zk, wit, wb, mask.r, mask.s, proof, ctx)
            print(f"   GPU proof == C oracle proof (bit-exact), pairing ok; the oracle took {cpu_s:.1f}s", flush=True)
        pk.destroy()


if __name__ == "__main__":
    main()
