"""Per-kernel timing of the hot-path ops at benchmark sizes (HIP events inside the library)."""
import argparse
import json
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nim_groth16_amd import Context  # noqa: E402
from tests.oracle_c import load_oracle  # noqa: E402
from tests import inputs as I  # noqa: E402
import numpy as np  # noqa: E402


def rand_fr_mont_bytes(n, seed):
    """fast: random 256-bit values reduced by masking the top bits (< 2^253 < r), used as Montgomery limbs."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 61) - 1)
    return a.tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--g2", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dist", choices=["uniform", "circom"], default="uniform",
                    help="circom: 40 %% zero, 30 %% one, 10 %% < 2^16, 20 %% uniform (SURVEY 8d config 2-ii)")
    ap.add_argument("--only", choices=["all", "ntt", "reg"], default="all",
                    help="ntt: NTT + quotient timings only; reg: registered-set MSMs (the prover's path) + NTT, without "
                         "the one-shot MSMs (clean per-kernel populations for rocprofv3 --pmc passes)")
    args = ap.parse_args()
    n = 1 << args.log2n
    orc = load_oracle()
    ctx = Context(0)
    ctx.selftest()
    if args.only == "ntt":
        return ntt_part(ctx, args, torch.frombuffer(bytearray(rand_fr_mont_bytes(4 * n, 2)), dtype=torch.uint8).cuda())
    t = time.time()
    kb = rand_fr_mont_bytes(n, 1)
    p1 = orc.fixed_base(1, kb)
    print(f"gen G1 points {time.time()-t:.1f}s", flush=True)
    sb = rand_fr_mont_bytes(n, 2)
    if args.dist == "circom":
        from oracle import bn254_ref as o
        rng = np.random.default_rng(3)
        u = rng.integers(0, 100, size=n)
        arr = np.frombuffer(bytearray(sb), dtype=np.uint64).reshape(n, 4).copy()
        one = np.frombuffer(o.fr_to_mont_bytes(1), dtype=np.uint64)
        arr[u < 40] = 0
        arr[(u >= 40) & (u < 70)] = one
        small = np.where((u >= 70) & (u < 80))[0]
        sm = [o.fr_to_mont_bytes(int(v)) for v in rng.integers(0, 1 << 16, size=len(small))]
        arr[small] = np.frombuffer(b"".join(sm), dtype=np.uint64).reshape(-1, 4)
        sb = arr.tobytes()
    d_s = torch.frombuffer(bytearray(sb), dtype=torch.uint8).cuda()
    d_p1 = torch.frombuffer(bytearray(p1), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    t = time.time()
    h1 = ctx.register_points(1, d_p1.data_ptr(), n, device=True)
    print(f"register G1 (tables) {time.time()-t:.3f}s", flush=True)
    reg_only = args.only == "reg"
    if not reg_only:
        assert ctx.msm_points(h1, d_s.data_ptr(), device=True) == ctx.msm(1, d_s.data_ptr(), d_p1.data_ptr(), n, device=True)
    for name, fn in [("msm_g1", lambda: ctx.msm(1, d_s.data_ptr(), d_p1.data_ptr(), n, device=True)),
                     ("msm_g1_registered", lambda: ctx.msm_points(h1, d_s.data_ptr(), device=True))][1 if reg_only else 0:]:
        fn()
        ctx.profile(True)
        ctx.profile_reset()
        t = time.time()
        for _ in range(args.reps):
            r = fn()
        dt = (time.time() - t) / args.reps
        rep = ctx.profile_report()
        ctx.profile(False)
        print(f"== {name} n=2^{args.log2n}: wall {dt*1e3:.3f} ms/call (with events)")
        for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"]):
            print(f"   {k:20s} {v['total_ms']/args.reps:9.3f} ms  ({v['calls']//args.reps} launches)")
        t = time.time()
        for _ in range(args.reps):
            r = fn()
        print(f"   wall without profiling: {(time.time()-t)/args.reps*1e3:.3f} ms/call")
    # correctness spot check vs oracle on a slice
    m = min(n, 1 << 12)
    if not reg_only:
        assert ctx.msm(1, sb[:32 * m], p1[:64 * m], m) == orc.msm(1, sb[:32 * m], p1[:64 * m])
    if args.g2:
        t = time.time()
        p2 = orc.fixed_base(2, kb)
        print(f"gen G2 points {time.time()-t:.1f}s", flush=True)
        d_p2 = torch.frombuffer(bytearray(p2), dtype=torch.uint8).cuda()
        t = time.time()
        h2 = ctx.register_points(2, d_p2.data_ptr(), n, device=True)
        print(f"register G2 (tables) {time.time()-t:.3f}s", flush=True)
        if not reg_only:
            assert ctx.msm_points(h2, d_s.data_ptr(), device=True) == ctx.msm(2, d_s.data_ptr(), d_p2.data_ptr(), n, device=True)
        fn = lambda: ctx.msm_points(h2, d_s.data_ptr(), device=True)
        fn()
        ctx.profile(True); ctx.profile_reset()
        for _ in range(args.reps):
            fn()
        rep = ctx.profile_report(); ctx.profile(False)
        print(f"== msm_g2 registered n=2^{args.log2n}")
        for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"]):
            print(f"   {k:20s} {v['total_ms']/args.reps:9.3f} ms  ({v['calls']//args.reps} launches)")
        t = time.time()
        for _ in range(args.reps):
            fn()
        print(f"   wall without profiling: {(time.time()-t)/args.reps*1e3:.3f} ms/call")
    ntt_part(ctx, args, torch.frombuffer(bytearray(rand_fr_mont_bytes(4 * n, 2)), dtype=torch.uint8).cuda())


def ntt_part(ctx, args, d_buf):
    """d_buf: 4n Fr elements in HBM (Az | Bz | Cz | out)"""
    n = 1 << args.log2n
    base = d_buf.data_ptr()
    d_y = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    cases = [("ntt fwd", lambda: ctx.ntt(base, args.log2n, False, dst=d_y.data_ptr(), device=True)),
             ("ntt inv", lambda: ctx.ntt(base, args.log2n, True, dst=d_y.data_ptr(), device=True)),
             ("quotient snarkjs (6 NTT + pointwise)",
              lambda: ctx.quotient(base, base + 32 * n, base + 64 * n, args.log2n, 1, out=base + 96 * n, device=True)),
             ("quotient jensgroth (7 NTT + pointwise)",
              lambda: ctx.quotient(base, base + 32 * n, base + 64 * n, args.log2n, 0, out=base + 96 * n, device=True))]
    for name, fn in cases:
        fn(); ctx.synchronize()
        ctx.profile(True); ctx.profile_reset()
        for _ in range(args.reps):
            fn()
        rep = ctx.profile_report(); ctx.profile(False)
        tot = sum(v["total_ms"] for v in rep.values()) / args.reps
        print(f"== {name} n=2^{args.log2n}: {tot:.3f} ms in kernels")
        for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"]):
            print(f"   {k:20s} {v['total_ms']/args.reps:9.3f} ms  ({v['calls']//args.reps} launches)")
        ctx.synchronize()
        t = time.time()
        for _ in range(10):
            fn()
        ctx.synchronize()
        print(f"   wall without profiling: {(time.time()-t)/10*1e3:.3f} ms/call")


if __name__ == "__main__":
    main()
