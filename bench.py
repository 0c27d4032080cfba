#!/usr/bin/env python3
"""Headline benchmark: Groth16 proofs/sec on the BN254 2^20-constraint synthetic circuit (BASELINE.json
configs[2]: full prove = buildABC + 6 NTTs + 4 G1 MSMs + 1 G2 MSM), inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  N > 1: launched by torch.distributed.run, one rank per GPU (RCCL).  --mode replica (default): every GPU
  proves its own proof per step, no data-path collective (weak scaling).  --mode shard: ONE proof per step,
  MSMs point-sharded over the GPUs + one all-gather of the 768-byte partial records (strong scaling).

Prints ONE JSON line (rank 0).  A "step" is one generateProofWithMask-equivalent (reference prover.nim:215-304).
Every run is gated on correctness: the GPU proof must equal the CPU oracle's proof bit for bit and satisfy the
pairing equation.  The oracle is used only for that check and for the `cpu_baseline` leg."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--log2n", type=int, default=20, help="domain size 2^log2n (constraints m = 2^log2n - 2)")
    ap.add_argument("--mode", choices=["replica", "shard"], default="replica")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend; gloo only for rehearsing the multi-process path on a one-GPU box "
                         "(all ranks then share device 0)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="replica mode: proofs in flight per GPU (each has its own context, streams and key copy); "
                         "the latency-bound tail of one proof overlaps the accumulation of the next")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    if args.backend == "gloo":
        local = 0                      # rehearsal: every rank computes on device 0
    torch.cuda.set_device(local)
    dist = None
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from nim_groth16_amd import Context, Mask, Witness, loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import SplitMix64, squaringChain
    from nim_groth16_amd.zkey_types import packCoeffs

    ctx = Context(local)
    ctx.selftest()
    n = 1 << args.log2n
    m = n - 2
    t0 = time.time()
    r1cs, wit = squaringChain(m, seed=4)
    rng = SplitMix64(5)
    tox = ToxicWaste(*[rng.fr() for _ in range(5)])
    zkey = fakeCircuitSetup(r1cs, tox, 1, ctx)           # scalar side on the host, every `y ** gen` on the GPU
    mrng = SplitMix64(6)
    mask = Mask(mrng.fr(), mrng.fr())
    wbytes = F.frSeqToMontBytes(wit)
    log(f"[bench] setup (synthetic circuit + fake trusted setup, domain 2^{args.log2n}): {time.time()-t0:.1f}s")

    t0 = time.time()
    shard = args.mode == "shard" and world > 1
    pkey = loadProvingKey(zkey, ctx, shard_index=rank if shard else 0, shard_count=world if shard else 1)
    d_w = torch.frombuffer(bytearray(wbytes), dtype=torch.uint8).cuda()
    rb, sb = F.frToMontBytes(mask.r), F.frToMontBytes(mask.s)
    torch.cuda.synchronize()
    log(f"[bench] key upload + window tables: {time.time()-t0:.1f}s")

    inflight = 1 if shard else max(1, args.inflight)
    lanes = [(ctx, pkey)]
    for _ in range(inflight - 1):
        c2 = Context(local)
        lanes.append((c2, loadProvingKey(zkey, c2)))
    torch.cuda.synchronize()

    if shard:
        from nim_groth16_amd._lib import PARTIALS_BYTES
        mine = torch.empty(PARTIALS_BYTES, dtype=torch.uint8, device="cuda")
        gathered = torch.empty(world * PARTIALS_BYTES, dtype=torch.uint8, device=coll_dev)

        def step(lane=0):
            pkey.prove_partials(d_w.data_ptr(), mont=True, device=True, out=mine.data_ptr())
            if coll_dev == "cuda":      # one RCCL all-gather of 768-byte records per proof
                dist.all_gather_into_tensor(gathered, mine)
                torch.cuda.current_stream().synchronize()
                return pkey.prove_combine(gathered.data_ptr(), world, rb, sb, device=True)
            dist.all_gather_into_tensor(gathered, mine.cpu())
            return pkey.prove_combine(gathered.numpy().tobytes(), world, rb, sb)
    else:
        def step(lane=0):
            return lanes[lane][1].prove(d_w.data_ptr(), mont=True, r=rb, s=sb, device=True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for c, _ in lanes:
            c.synchronize()

    import threading
    last = [None] * inflight

    def run(count):
        """`count` proofs, `inflight` at a time: worker i proves steps i, i+inflight, ... on its own context"""
        if inflight == 1:
            for _ in range(count):
                last[0] = step()
            return
        def work(i):
            for _ in range(i, count, inflight):
                last[i] = step(i)
        th = [threading.Thread(target=work, args=(i,)) for i in range(inflight)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    run(max(args.warmup, inflight))
    barrier()
    # HIP-event timing of the dominant kernels (bucket accumulation) runs INSIDE the timed region, on this rank's
    # first context; events are recorded on the stream each kernel is launched on.  (Events around all ~130 launches
    # of a proof cost ~3 % of throughput, so the full per-kernel breakdown is taken on extra steps afterwards.)
    ctx.profile(2 if rank == 0 else 0)
    ctx.profile_reset()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    rep = ctx.profile_report() if rank == 0 else {}
    ctx.profile(False)
    proof = last[0]
    assert all(p == proof for p in last if p is not None), "in-flight lanes disagree"
    # single-proof latency (one proof in flight), reported next to the throughput
    t1 = time.perf_counter()
    for _ in range(3):
        step()
    lat_ms = (time.perf_counter() - t1) / 3 * 1e3
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    proofs = args.steps * (1 if shard or world == 1 else world)
    value = proofs / dt

    # ---- roofline of the dominant kernel of the step (from the events of the timed region) -------------------
    roof, extra = None, {}
    reps = 3
    if rank == 0 or shard:            # in shard mode a step contains a collective: every rank must take part
        ctx.profile(1 if rank == 0 else 0)
        ctx.profile_reset()
        for _ in range(reps):
            step()
        rep_all = ctx.profile_report() if rank == 0 else {}
        ctx.profile(False)
    if rank == 0:
        kern = {k: v["total_ms"] / v["calls"] for k, v in rep.items()}
        kern_all = {k: v["total_ms"] / v["calls"] for k, v in rep_all.items()}
        calls = {k: v["calls"] // reps for k, v in rep_all.items()}
        dom = max(rep, key=lambda k: rep[k]["total_ms"])
        nsh = (n // world) if shard else n
        # algorithmic bytes of one launch: a G1 MSM reads 32+64 B per pair, a G2 MSM 32+128 B
        per_pair = {"g1": 96, "g2": 160}
        alg = per_pair["g2" if dom.endswith("g2") else "g1"] * nsh if dom.startswith("msm_") else 64 * n
        if dom in ("msm_count", "msm_scatter"):
            alg = 32 * nsh
        achieved = alg / (kern[dom] * 1e-3) / 1e9
        # measured HBM bytes per launch of that kernel (rocprofv3 PMC passes, profiles/r01_pmc_hbm_traffic_2p20.json)
        traffic = None
        try:
            if args.log2n == 20 and not shard:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic_2p20.json")))
                traffic = pm["kernels"][dom]["hbm_bytes_per_launch_raw"]
        except Exception:
            traffic = None
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 6), "traffic": traffic,
                "avg_launch_ms": round(kern[dom], 4), "algorithmic_bytes_per_launch": alg,
                "note": "MSM is integer-ALU-bound (254-bit Montgomery madds), not HBM-bound: see DESIGN.md"}
        # all kernels of one proof (separate steps, one proof in flight, every launch bracketed by events)
        extra["kernel_ms_per_proof"] = {k: round(kern_all[k] * calls[k], 4) for k in sorted(kern_all)}
        # G1-adds/sec: one stand-alone registered G1 MSM (witness x pointsA1), all phases, HIP-event timed
        nsh = n
        hA = ctx.register_points(1, zkey.pPoints.pointsA1, zkey.header.nvars)
        c, W = hA.info()
        ctx.msm_points(hA, d_w.data_ptr(), device=True)
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(reps):
            ctx.msm_points(hA, d_w.data_ptr(), device=True)
        rep1 = ctx.profile_report()
        ctx.profile(False)
        hA.release()
        g1 = sum(v["total_ms"] for v in rep1.values()) / reps
        adds = nsh * W + 2 * (1 << (c - 1))     # bucket additions + running-sum reduction (one merged bucket set)
        extra["msm_g1_adds_per_sec"] = round(adds / (g1 * 1e-3), 1)
        extra["msm_g1_pairs_per_sec"] = round(nsh / (g1 * 1e-3), 1)
        extra["msm_g1_ms"] = round(g1, 4)
        if dom in rep1:     # the same kernel with the GPU to itself (no other lane / proof in flight)
            iso = rep1[dom]["total_ms"] / rep1[dom]["calls"]
            roof["isolated_launch_ms"] = round(iso, 4)
            roof["achieved_isolated"] = round(alg / (iso * 1e-3) / 1e9, 3)
        roof["note"] = ("MSM is integer-ALU-bound (254-bit Montgomery madds), not HBM-bound: see DESIGN.md.  avg_launch_ms is "
                        "the HIP-event duration inside the timed region, where this kernel shares the GPU with the other "
                        "MSM lanes and in-flight proofs")
        extra["msm_window_bits"] = c
        extra["msm_tables"] = W

    # ---- correctness gate + CPU baseline (oracle = checker / baseline only) ----------------------------------
    cpu = None
    if rank == 0:
        from oracle import bn254_ref as o
        from tests.oracle_c import load_oracle
        orc = load_oracle()
        pts, hdr = zkey.pPoints, zkey.header
        packed = packCoeffs(zkey.coeffs)
        t0 = time.perf_counter()
        Az, Bz, Cz = orc.build_abc(packed, wbytes, args.log2n)
        qs = orc.quotient_snarkjs(Az, Bz, Cz, args.log2n, parallel=True)
        mA = orc.msm(1, wbytes, pts.pointsA1)
        mB1 = orc.msm(1, wbytes, pts.pointsB1)
        mB2 = orc.msm(2, wbytes, pts.pointsB2)
        mH = orc.msm(1, qs, pts.pointsH1)
        mC = orc.msm(1, wbytes[32 * (hdr.npubs + 1):], pts.pointsC1)
        cpu_s = time.perf_counter() - t0
        it = iter([o.g1_from_bytes(mA), o.g1_from_bytes(mB1), o.g2_from_bytes(mB2), o.g1_from_bytes(mH),
                   o.g1_from_bytes(mC)])
        oz = o.ZKey()
        oz.flavour, oz.nvars, oz.npubs, oz.domainSize = o.SNARKJS, hdr.nvars, hdr.npubs, hdr.domainSize
        sp = zkey.specPoints
        oz.alpha1, oz.beta1, oz.delta1 = (o.g1_from_bytes(x) for x in (sp.alpha1, sp.beta1, sp.delta1))
        oz.beta2, oz.gamma2, oz.delta2 = (o.g2_from_bytes(x) for x in (sp.beta2, sp.gamma2, sp.delta2))
        oz.pointsIC = [o.g1_from_bytes(zkey.pointsIC[i:i + 64]) for i in range(0, len(zkey.pointsIC), 64)]
        dummy = [None] * hdr.nvars
        oz.pointsA1 = oz.pointsB1 = oz.pointsB2 = dummy
        oz.pointsC1, oz.pointsH1, oz.coeffs = [None] * (hdr.nvars - hdr.npubs - 1), [None] * n, []
        ref = o.generate_proof_with_mask(oz, wit, mask.r, mask.s, msm_g1=lambda c_, p_: next(it),
                                         msm_g2=lambda c_, p_: next(it), quotient=lambda *a: [0] * n)
        got = (o.g1_from_bytes(proof[0]), o.g2_from_bytes(proof[1]), o.g1_from_bytes(proof[2]))
        if got != (ref.pi_a, ref.pi_b, ref.pi_c):
            raise SystemExit("FAIL: GPU proof differs from the CPU oracle's proof")
        if not o.verify_proof(oz, ref):
            raise SystemExit("FAIL: proof does not satisfy the pairing equation")
        from nim_groth16_amd import Proof, extractVKey, verifyProof
        pio = wbytes[:32 * (hdr.npubs + 1)]      # Proof.publicIO = witness[0..npubs] (prover.nim:238-240)
        if not verifyProof(extractVKey(zkey), Proof(pio, proof[0], proof[1], proof[2]), ctx):
            raise SystemExit("FAIL: the GPU verifier rejects the proof")
        log(f"[bench] correctness gate passed: GPU proof == CPU oracle proof (bit-exact), pairing check ok "
            f"(oracle and GPU verifier)")
        if not args.no_cpu_baseline and world == 1:     # reported at N = 1 only
            cpu = {"value": round(1.0 / cpu_s, 5), "unit": "proofs/s", "cores": orc.cores(), "kind": "port",
                   "sample": f"1 full proof (buildABC + 6 NTT + 4 G1 MSM + 1 G2 MSM), domain 2^{args.log2n}, "
                             f"{cpu_s:.1f}s; C restatement, NOT constantine"}

    if rank == 0:
        out = {
            "metric": "proofs/sec (BN254 Groth16, 2^20-constraint circuit)" if args.log2n == 20 else
                      f"proofs/sec (BN254 Groth16, 2^{args.log2n} domain)",
            "value": round(value, 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "u32 (254-bit Montgomery: 8x32-bit limbs; 9x29-bit limbs in the bucket accumulation)",
            "data": "synthetic",
            "config": {"workload": f"BN254 2^{args.log2n}-constraint synthetic R1CS (squaring chain, m=2^{args.log2n}-2), "
                                   "full prove: buildABC + 6 NTT + 4 G1 MSM + 1 G2 MSM, snarkjs flavour, 1 proof/step/GPU"
                                   f" ({inflight} proofs in flight per GPU)"
                                   if not shard else
                                   f"BN254 2^{args.log2n}-constraint synthetic R1CS, ONE proof per step, MSMs point-sharded "
                                   f"over {world} GPUs + all-gather of partials",
                       "mode": "shard" if shard else "replica", "nvars": zkey.header.nvars,
                       "domain_log2": args.log2n, "inputs": "witness + proving key resident in HBM"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        extra["proof_latency_ms_single_in_flight"] = round(lat_ms, 3)
        extra["proofs_in_flight_per_gpu"] = inflight
        out.update(extra)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
