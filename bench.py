#!/usr/bin/env python3
"""Headline benchmark: Groth16 proofs/sec on the BN254 2^20-constraint synthetic circuit (BASELINE.json
configs[2]: full prove = buildABC + 6 NTTs + 4 G1 MSMs + 1 G2 MSM).

  python bench.py --gpus N --steps K --warmup W

  N > 1 from a bare shell: bench.py starts its own N ranks (child `python -m torch.distributed.run`, one rank per
  GPU over RCCL) BEFORE anything touches a GPU, relays rank 0's JSON line and exits with the children's code.  Under
  an existing torch.distributed.run launch (WORLD_SIZE set) it is a rank; WORLD_SIZE != --gpus is an error.
  --mode replica (default): every GPU proves its own proofs, no data-path collective (weak scaling).
  --mode shard: ONE proof per step, MSMs point-sharded over the GPUs + one all-gather of the 768-byte partial
  records (strong scaling).

A "step" is one generateProofWithMask-equivalent (reference prover.nim:215-304) on a HOST witness: as in the
reference (`generateProof(zkey, wtns)` takes a parsed .wtns, prover.nim:215-240) the proving key is resident (parsed
once per circuit, files/zkey.nim:241-245) and the witness arrives per proof -- here from pinned host memory in the
.wtns layout (standard form, files/witness.nim:14), rotating over several DISTINCT satisfying witnesses of the same
circuit, so `value` includes the PCIe transfer.  The HBM-resident figure is reported next to it
(`value_witness_in_hbm`).  The in-flight proofs of one GPU share ONE resident key.

Prints ONE JSON line (rank 0).  Every run is gated on correctness: for every distinct witness the GPU proof must
equal the CPU oracle's proof bit for bit and satisfy the pairing equation (tests/parity.py).  The oracle is used only
for that check and for the `cpu_baseline` leg."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); the 18 streams of three in-flight
# proofs do measurably better on 8 (same-box A/B, profiles/r03_ab_hwqueues.txt: 114.8 -> 116.4 proofs/s; 6 and 16 are
# worse).  Read by the HIP runtime when it initialises, so it is set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

GATHER_CEILING_GB_S = 2200.0     # measured ceiling of random table gathers (profiles/r03_ubench_affine_g2.txt)
NWITNESS = 3          # distinct satisfying witnesses the timed steps rotate over (w_0 = 3, 5, 7; same k_i, same key)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=288,
                    help="timed proofs (default 288: a ~2.5-s timed region at 2^20; round 2's 96 steps = 0.9 s were within box noise)")
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--repeats", type=int, default=0,
                    help="the timed region (exactly --steps proofs, barriers on both sides) is run this many times; "
                         "`value` / `ms_per_step` are the median run, every run is listed in `value_runs`.  Default: 5, "
                         "or 11 for regions of fewer than 96 steps (a 20-step region is 0.17 s: single ones scatter by "
                         "+-4 %, medians of five by +-1.7 % on the noisiest box, profiles/r04_bench_repeat.txt)")
    ap.add_argument("--log2n", type=int, default=20, help="domain size 2^log2n (constraints m = 2^log2n - 2)")
    ap.add_argument("--mode", choices=["replica", "shard"], default="replica")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend; gloo only for rehearsing the multi-process path on a one-GPU box "
                         "(all ranks then share device 0)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="replica mode: proofs in flight per GPU (one context each: private streams and workspaces, "
                         "ONE shared resident key); the latency-bound tail of one proof overlaps the next one's "
                         "accumulation")
    ap.add_argument("--witness", choices=["host", "hbm"], default="host",
                    help="where the timed steps take the witness from (host = pinned memory, .wtns layout)")
    ap.add_argument("--quotient", choices=["tasks", "replicated"], default="tasks",
                    help="shard mode: the three coset pipelines of the quotient on three different ranks + three "
                         "scatters of the slices (tasks), or recomputed by every rank (replicated)")
    ap.add_argument("--no-poseidon-shape", action="store_true",
                    help="skip the extra key value_poseidon_shape (N = 1, replica mode: a second key, a Poseidon-shaped "
                         "Merkle circuit of the same domain, ~15 s of setup + one more oracle proof)")
    ap.add_argument("--inject-failure", default="", metavar="RANK[:AFTER]",
                    help="test hook: that rank raises after AFTER proofs (dry run: before its first collective)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / reduction plumbing only (gloo, no GPU, no proofs): what the CPU test "
                         "of the multi-rank launch path runs; prints a line with value 0 and dry_run true")
    return ap.parse_args(argv)


def launch_command(args, argv, port):
    """the child command for `bench.py --gpus N` from a bare shell (N > 1): the driver's own launch line"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """Start the N ranks as CHILD processes (never exec: this process has not touched a GPU, and must not replace
    itself either), relay their output, return their exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: needed by RCCL on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launch_command(args, argv, port)
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def lib_sha16():
    """identity of the kernels the static roofline inputs must belong to (nim_groth16_amd/_lib.py)"""
    from nim_groth16_amd._lib import device_code_sha16
    return device_code_sha16()


def build_inputs(args, ctx, rank, world, dist):
    """The circuit, its key and NWITNESS satisfying witnesses.  Rank 0 alone runs the ~25 s of Python big-integer
    setup (and the fixed-base multiplications on its GPU); the other ranks of a multi-GPU launch load what it wrote
    (one pickle in /dev/shm) instead of repeating it."""
    import pickle
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import SplitMix64, squaringChain
    n = 1 << args.log2n
    m = n - 2
    import shutil
    import tempfile
    path = None
    if rank == 0:
        r1cs, wit0 = squaringChain(m, seed=4)
        # more satisfying witnesses of the SAME circuit: the chain constants k_i (which are what the key depends on)
        # stay, the free input w_0 changes, and with it every wire
        wits = [wit0] + [squaringChain(m, seed=4, w0=w0)[1] for w0 in (5, 7, 11, 13)[:NWITNESS - 1]]
        rng = SplitMix64(5)
        tox = ToxicWaste(*[rng.fr() for _ in range(5)])
        zkey = fakeCircuitSetup(r1cs, tox, 1, ctx)           # scalar side on the host, every `y ** gen` on the GPU
        if world > 1:
            # a directory only this user can enter (mkdtemp: mode 0700, unpredictable name), its path handed to the
            # other ranks through the process group -- not a fixed name in world-writable /dev/shm
            shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
            path = os.path.join(tempfile.mkdtemp(prefix="g16bench_", dir=shm), "key.pkl")
            with open(path, "wb") as f:
                pickle.dump((zkey, wits), f, protocol=pickle.HIGHEST_PROTOCOL)
    if world > 1:
        box = [path]
        dist.broadcast_object_list(box, src=0)
        if rank != 0:
            with open(box[0], "rb") as f:
                zkey, wits = pickle.load(f)
        dist.barrier()
        if rank == 0:
            shutil.rmtree(os.path.dirname(path), ignore_errors=True)
    return zkey, wits


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")

    import torch
    if args.dry_run:
        return dry_run(args, rank, world)
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

    # A failure on ANY rank (a worker error, a library error, a proof that differs) ends THAT rank at once: `FAIL ...` on
    # stderr, exit code 1, no further collective.  torch.distributed.run then tears the other ranks down and the launch
    # returns non-zero (the `quit()` convention of cli/cli_main.nim:187-222).  Rounds 3-4 shared the failure through an
    # all-reduced flag instead -- but the healthy ranks are by then inside a DIFFERENT collective (the next proof's
    # scatter / all-gather, or a barrier), the flag pairs with a mismatched operation and every rank waits forever
    # (profiles/r05_failpath_old_one_rank_fails_hangs.err.txt).
    state = {}
    try:
        measure(args, rank, world, local, dist, coll_dev, state)
    except BaseException as e:       # noqa: BLE001
        fail_fast(e, rank, world)
    if dist is not None:
        dist.destroy_process_group()
    # The oracle gate and the CPU baseline run AFTER the process group is gone: 3 x ~14 s of CPU work on rank 0
    # during which no other rank idles inside a collective.
    if rank == 0:
        finish_rank0(args, world, state)


def fail_fast(e, rank, world):
    """`FAIL (rank r): ...` + the traceback on stderr, then leave with code 1 WITHOUT interpreter teardown when other
    ranks exist: destructors of contexts with proofs in flight, or of a process group with collectives pending, have
    nothing to wait for that will ever arrive."""
    import traceback
    msg = str(e) if isinstance(e, SystemExit) else repr(e)
    if not msg.startswith("FAIL"):
        msg = "FAIL: " + msg
    sys.stderr.write(f"{msg}  [rank {rank} of {world}]\n")
    if not isinstance(e, SystemExit):
        traceback.print_exception(type(e), e, e.__traceback__, file=sys.stderr)
    sys.stderr.flush()
    sys.stdout.flush()
    if world > 1:
        os._exit(1)
    raise SystemExit(msg)


def injected_failure(args, rank, done):
    """--inject-failure RANK[:AFTER]: that rank fails once `AFTER` (default 2) of its timed-path proofs are done (tests of
    the failure path; `done` = None in the dry run: fail before the first collective)"""
    if not args.inject_failure:
        return
    r, _, after = args.inject_failure.partition(":")
    if int(r) == rank and (done is None or done >= int(after or 2)):
        raise RuntimeError(f"injected failure on rank {rank}")


def measure(args, rank, world, local, dist, coll_dev, state):
    import threading
    import torch
    from nim_groth16_amd import Context, Mask, loadProvingKey
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.synthetic import SplitMix64

    ctx = Context(local)
    ctx.selftest()
    n = 1 << args.log2n
    t0 = time.time()
    zkey, wits = build_inputs(args, ctx, rank, world, dist)
    mrng = SplitMix64(6)
    mask = Mask(mrng.fr(), mrng.fr())
    log(f"[bench] setup (synthetic circuit, {NWITNESS} witnesses, fake trusted setup, domain 2^{args.log2n}): "
        f"{time.time()-t0:.1f}s")

    t0 = time.time()
    shard = args.mode == "shard" and world > 1
    pkey = loadProvingKey(zkey, ctx, shard_index=rank if shard else 0, shard_count=world if shard else 1)
    # the witnesses as a .wtns would deliver them: standard-form bytes in (pinned) host memory
    h_w = []
    for w in wits:
        t = torch.empty(32 * len(w), dtype=torch.uint8).pin_memory()
        t.copy_(torch.frombuffer(bytearray(F.frSeqToStdBytes(w)), dtype=torch.uint8))
        h_w.append(t)
    d_w = [t.cuda() for t in h_w]                          # HBM-resident copies for the `value_witness_in_hbm` leg
    rb, sb = F.frToMontBytes(mask.r), F.frToMontBytes(mask.s)
    torch.cuda.synchronize()
    log(f"[bench] key upload + window tables: {time.time()-t0:.1f}s")

    inflight = max(1, args.inflight)
    proofs = [None] * NWITNESS         # last proof seen per witness
    plock = threading.Lock()
    errors = []                        # worker threads cannot end the process: failures are collected and re-raised

    def keep(i, p):
        with plock:
            k = i % NWITNESS
            if proofs[k] is not None and proofs[k] != p:
                errors.append(f"two proofs of witness {k} differ (step {i})")
            proofs[k] = p

    if shard:
        # one proof over all ranks (nim_groth16_amd/distributed.py): sharded MSMs, the three coset pipelines of the
        # quotient on three different ranks (--quotient tasks) or replicated on all of them (--quotient replicated),
        # one all-gather of the 768-byte partial records; `--inflight` sharded proofs overlap (a context each)
        from nim_groth16_amd.distributed import ShardedProver
        inflight = min(inflight, 2) if args.inflight == 3 else inflight      # default depth of the shard pipeline: 2
        sp = ShardedProver(zkey, rank, world, ctx=ctx, quotient=args.quotient, pkey=pkey, depth=inflight)
        ctxs = [ctx]

        def step(i, lane=0, hbm=False):
            w = (d_w if hbm else h_w)[i % NWITNESS]
            return sp.prove_raw(w.data_ptr(), False, rb, sb, device=hbm)

        def run(count, hbm=False):
            """`count` sharded proofs through the pipeline (submit / collect, `inflight` in flight)"""
            done = 0
            for i in range(count):
                w = (d_w if hbm else h_w)[i % NWITNESS]
                p = sp.submit(w.data_ptr(), False, rb, sb, device=hbm)
                if p is not None:
                    keep(done, p)
                    done += 1
                    injected_failure(args, rank, done)
            for p in sp.collect():
                keep(done, p)
                done += 1
            if errors:
                raise SystemExit("FAIL: " + "; ".join(errors))
    else:
        ctxs = [ctx] + [Context(local) for _ in range(inflight - 1)]      # ONE key, `inflight` contexts

        def step(i, lane=0, hbm=False, pk=None, ws=None):
            ws = ws or (d_w if hbm else h_w)
            return (pk or pkey).prove(ws[i % len(ws)].data_ptr(), mont=False, r=rb, s=sb, device=hbm, ctx=ctxs[lane])

        def run(count, hbm=False, pk=None, ws=None, sink=keep):
            """`count` proofs, `inflight` at a time: worker j proves steps j, j+inflight, ... on its own context"""
            def work(j):
                try:
                    torch.cuda.set_device(local)   # a new host thread starts on device 0 (the library sets it per call too)
                    for i in range(j, count, inflight):
                        sink(i, step(i, j, hbm, pk, ws))
                        injected_failure(args, rank, i + 1)
                except BaseException as e:      # a G16Error in a worker must fail the run, not just end the thread
                    with plock:
                        errors.append(f"worker {j}: {e!r}")
            th = [threading.Thread(target=work, args=(j,)) for j in range(inflight)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if errors:
                raise SystemExit("FAIL: " + "; ".join(errors))
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for c in ctxs:
            c.synchronize()

    def allmax(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    run(max(args.warmup, inflight, NWITNESS), hbm=args.witness == "hbm")
    barrier()
    # HIP events around the bucket-accumulation kernels run INSIDE the timed region, on this rank's first context
    # (recorded on the stream each kernel is launched on): the CONTENDED duration of the dominant kernel, reported
    # next to the uncontended one below.  (Events around all ~110 launches of a proof would cost ~3 % of throughput.)
    ctx.profile(2 if rank == 0 else 0)
    ctx.profile_reset()
    runs = []
    repeats = args.repeats if args.repeats > 0 else (5 if args.steps >= 96 else 11)
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        run(args.steps, hbm=args.witness == "hbm")
        barrier()
        runs.append(allmax(time.perf_counter() - t0))
    rep = ctx.profile_report() if rank == 0 else {}
    ctx.profile(False)
    dt = sorted(runs)[len(runs) // 2]          # the median run
    proofs_done = args.steps * (1 if shard or world == 1 else world)
    value = proofs_done / dt
    # the other witness placement, same protocol (barrier, K' steps, barrier, max over ranks), reported as an extra key
    k2 = max(inflight * NWITNESS, min(args.steps, 48))
    barrier()
    t1 = time.perf_counter()
    run(k2, hbm=args.witness != "hbm")
    barrier()
    dt2 = allmax(time.perf_counter() - t1)
    value_other = k2 * (1 if shard or world == 1 else world) / dt2
    # single-proof latency (one proof in flight, host witness), reported next to the throughput
    barrier()
    t1 = time.perf_counter()
    for i in range(3):
        step(i)
    lat_ms = (time.perf_counter() - t1) / 3 * 1e3
    barrier()
    if shard:
        sp.close()
    if not shard and world == 1 and not args.no_poseidon_shape:
        # an EXTRA measurement: whatever goes wrong in setting it up (memory, an oversized domain) must not cost the run
        # its headline; a wrong Poseidon-shape PROOF still fails the run (finish_rank0's gate)
        try:
            state["poseidon"] = poseidon_shape(args, ctx, run, step, barrier, rb, sb)
        except (Exception, SystemExit) as e:        # noqa: BLE001  (run() reports worker errors as SystemExit)
            if "differ" in str(e):                  # two proofs of one witness that differ: a correctness failure
                raise
            log(f"[bench] value_poseidon_shape skipped: {e!r}")
            state["poseidon_error"] = repr(e)
            errors.clear()

    state.update(zkey=zkey, wits=wits, mask=mask, proofs=proofs, ctx=ctx, value=value, dt=dt, value_other=value_other,
                 lat_ms=lat_ms, inflight=inflight, shard=shard, value_runs=[round(proofs_done / t, 4) for t in runs])
    if rank != 0:
        return
    # ---- roofline of the dominant kernel (rank 0; nothing else runs on this GPU from here on) -------------------------
    extra = {}
    kern = {k: v["total_ms"] / v["calls"] for k, v in rep.items()}
    dom = max(rep, key=lambda k: rep[k]["total_ms"])
    nsh = (n // world) if shard else n
    # algorithmic bytes of one launch: a G1 MSM reads 32+64 B per pair, a G2 MSM 32+128 B
    alg = (160 if dom.endswith("g2") else 96) * nsh
    # The same kernel with the GPU to itself: `reps` stand-alone registered MSMs over the same point array and the
    # same witness, every launch bracketed by HIP events on its stream and stamping the shader clock from inside.
    reps = 5

    def isolated(group):
        pts = zkey.pPoints.pointsB2 if group == 2 else zkey.pPoints.pointsA1
        h = ctx.register_points(group, pts, zkey.header.nvars)
        cW = h.info()
        ctx.msm_points(h, d_w[0].data_ptr(), mont=False, device=True)
        ctx.profile(True)
        ctx.profile_reset()
        ctx.profile_clock()                              # reset the in-kernel clock sums
        t1 = time.perf_counter()
        for _ in range(reps):
            ctx.msm_points(h, d_w[0].data_ptr(), mont=False, device=True)
        wall = (time.perf_counter() - t1) / reps * 1e3
        ghz = ctx.profile_clock()                        # s_memtime / s_memrealtime inside those launches
        r = ctx.profile_report()
        ctx.profile(False)
        h.release()
        return r, wall, ghz, cW

    grp = 2 if dom.endswith("g2") else 1
    rep1, msm_wall, clock_ghz, (c, W) = isolated(grp)
    rep_other = isolated(3 - grp)[0] if not shard else None
    iso = rep1[dom]["total_ms"] / rep1[dom]["calls"]
    achieved = alg / (iso * 1e-3) / 1e9
    static = static_inputs(args, shard, dom)
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
            "frac": round(achieved / 8000.0, 6), "traffic": static["traffic"],
            "avg_launch_ms": round(iso, 4), "launches_timed": rep1[dom]["calls"],
            "algorithmic_bytes_per_launch": alg,
            "contended_launch_ms_in_timed_region": round(kern[dom], 4),
            "inputs_from": static["traffic_from"],
            "note": ("avg_launch_ms: HIP events on the launch stream, this kernel ALONE on the GPU (stand-alone MSMs of "
                     "the same points and witness right after the timed region); contended_...: the same events inside "
                     "the timed region, where ~15 streams share the CUs -- overlapped wall time, not a per-step cost. "
                     "MSM is integer-ALU-bound (254-bit Montgomery madds), not HBM-bound: see roofline_valu")}
    # the next roof: the window tables are read by random 64 / 128-byte gathers, whose measured ceiling on this chip is
    # far below the streaming peak (tools/ubench_affine_g2: ~2.2 TB/s, profiles/r03_ubench_affine_g2.txt)
    extra["roofline_gather"] = {
        "kernel": dom, "hbm_bytes_per_launch_by_counters": static["traffic"], "avg_launch_ms": round(iso, 4),
        "achieved_gb_s": None if static["traffic"] is None else round(static["traffic"] / (iso * 1e-3) / 1e9, 1),
        "ceiling_gb_s": GATHER_CEILING_GB_S, "ceiling_from": "profiles/r03_ubench_affine_g2.txt (random 64-128-byte gathers "
        "from a 1.7-GB table, 1-2 waves/SIMD)",
        "frac": None if static["traffic"] is None else round(static["traffic"] / (iso * 1e-3) / 1e9 / GATHER_CEILING_GB_S, 4),
        "inputs_from": static["traffic_from"]}
    # how much of a step is bucket accumulation: the isolated accumulate launches of one proof (4 G1 + 1 G2) over the
    # measured step -- the rest of the step is tails, sorts, NTTs and whatever the in-flight proofs fail to overlap
    if rep_other is not None:
        both = {**{k: v["total_ms"] / v["calls"] for k, v in rep1.items()},
                **{k: v["total_ms"] / v["calls"] for k, v in rep_other.items()}}
        if "msm_accum_g1" in both and "msm_accum_g2" in both:
            acc_ms = 4 * both["msm_accum_g1"] + both["msm_accum_g2"]
            extra["accum_ms_per_proof"] = round(acc_ms, 4)
            extra["overlap_efficiency"] = overlap_efficiency(acc_ms, dt, args.steps)
    g1 = sum(v["total_ms"] for v in rep1.values()) / reps
    if grp == 1:
        nwin = 254 // c + 1                   # W tables = nwin windows x (1 or 2) multiplier tables
        h = 1 << (c - 1)
        nbuckets = h if W == nwin else h // 2 + h // 8 + h // 32 + h // 64      # class bucket set with two tables
        adds = n * nwin + 2 * nbuckets        # bucket additions + running-sum reduction (one merged bucket set)
        extra["msm_buckets"] = nbuckets
        extra["msm_g1_adds_per_sec"] = round(adds / (g1 * 1e-3), 1)
        extra["msm_g1_pairs_per_sec"] = round(n / (g1 * 1e-3), 1)
        extra["msm_g1_ms"] = round(g1, 4)
    extra["msm_window_bits"] = c
    extra["msm_tables"] = W
    # stand-alone kernel times of that MSM (one MSM alone on the GPU; a proof runs five of them on five streams)
    extra["kernel_ms_standalone_msm_g%d" % grp] = {k: round(v["total_ms"] / reps, 4) for k, v in sorted(rep1.items())}
    extra["standalone_msm_wall_ms"] = round(msm_wall, 4)
    extra["roofline_valu"] = valu_roofline(dom, iso, clock_ghz, static)
    state.update(roof=roof, extra=extra)


def overlap_efficiency(accum_ms_per_proof, region_s, steps):
    """isolated accumulate time of one proof over the time THIS GPU spends per proof.  In replica mode every rank proves
    `steps` proofs in the region, in shard mode all ranks work on each of the `steps` proofs: either way a GPU's time
    per proof is region / steps -- the same expression as `ms_per_step`, for any world size."""
    return round(accum_ms_per_proof / (region_s / steps * 1e3), 4)


def poseidon_shape(args, ctx, run, step, barrier, rb, sb):
    """The same timed protocol on BASELINE config 5's workload SHAPE: a Poseidon-shaped Merkle-inclusion circuit
    (nim_groth16_amd/synthetic.py: rows of 1..25 terms, ncoeffs ~ 12 n, a third of the wires absent from B -- not a
    circom artefact: none can be produced here), same domain, same toxic waste, its own key.  Reported as the extra key
    `value_poseidon_shape`; `value` is untouched."""
    import torch
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd import loadProvingKey
    from nim_groth16_amd.fake_setup import ToxicWaste, fakeCircuitSetup
    from nim_groth16_amd.synthetic import SplitMix64, poseidonMerkle
    t0 = time.time()
    r1cs, wit = poseidonMerkle(args.log2n, seed=4)
    rng = SplitMix64(5)
    zk = fakeCircuitSetup(r1cs, ToxicWaste(*[rng.fr() for _ in range(5)]), 1, ctx)
    pk = loadProvingKey(zk, ctx)
    ws = F.frSeqToStdBytes(wit)
    hw = torch.empty(len(ws), dtype=torch.uint8).pin_memory()
    hw.copy_(torch.frombuffer(bytearray(ws), dtype=torch.uint8))
    log(f"[bench] Poseidon-shaped circuit, setup, key upload: {time.time()-t0:.1f}s")
    last = [None]

    def sink(i, p):
        if last[0] is not None and last[0] != p:
            raise RuntimeError("two proofs of the Poseidon-shaped witness differ")
        last[0] = p
    steps = max(args.steps, 48)
    run(12, pk=pk, ws=[hw], sink=sink)
    vals = []
    for _ in range(3):
        barrier()
        t1 = time.perf_counter()
        run(steps, pk=pk, ws=[hw], sink=sink)
        barrier()
        vals.append(steps / (time.perf_counter() - t1))
    t1 = time.perf_counter()
    for i in range(3):
        step(i, pk=pk, ws=[hw])
    lat = (time.perf_counter() - t1) / 3 * 1e3
    barrier()
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(5):
        pk.build_abc(ws, mont=False)
    rep = ctx.profile_report()
    ctx.profile(False)
    info, inf = pk.abc_info(), pk.inf_counts()
    pk.destroy()
    return {"zkey": zk, "wit": wit, "proof": last[0],
            "line": {"value": round(sorted(vals)[1], 4), "unit": "proofs/s", "runs": [round(v, 4) for v in vals],
                     "steps": steps, "proof_latency_ms_single_in_flight": round(lat, 3),
                     "circuit": "Poseidon-shaped Merkle inclusion (seeded constants, NOT circomlib-compatible), "
                                f"{r1cs.depth} levels, {r1cs.nConstraints} constraints, domain 2^{args.log2n}",
                     "nvars": zk.header.nvars, "ncoeffs": info["ncoeffs"],
                     "ncoeffs_per_domain_row": round(info["ncoeffs"] / zk.header.domainSize, 3),
                     "coefficient_dictionary_values": info["dict_values"], "rows_by_terms": info["rows_by_terms"],
                     "points_at_infinity": {k: inf[k] for k in ("A1", "B1", "B2", "C1", "H1")},
                     "abc_spmv_ms_standalone": round(rep["abc_spmv"]["total_ms"] / rep["abc_spmv"]["calls"], 4),
                     "abc_cz_ms_standalone": round(rep["abc_cz"]["total_ms"] / rep["abc_cz"]["calls"], 4)}}


def static_inputs(args, shard, dom):
    """Per-launch quantities that only a rocprofv3 counter pass can deliver (HBM bytes, VALU wave-instructions): read
    from the committed files of THIS build -- every file names the hash of the device code (the .hip_fatbin section of
    libg16hip.so) it was measured on, and anything measured on other kernels is dropped (null) rather than reported as if it belonged to this run."""
    out = {"traffic": None, "traffic_from": None, "valu": None, "valu_from": None}
    if args.log2n != 20 or shard:
        return out
    sha = lib_sha16()
    for key, names in (("traffic", ("r05_pmc_hbm_traffic_2p20.json", "r04_pmc_hbm_traffic_2p20.json")),
                       ("valu", ("r05_valu_roofline_inputs.json", "r04_valu_roofline_inputs.json"))):
        for name in names:
            path = os.path.join(ROOT, "profiles", name)
            try:
                d = json.load(open(path))
                if d.get("kernels_sha16") != sha:
                    out[key + "_from"] = {"file": "profiles/" + name, "dropped": "measured on other kernels "
                                          f"({d.get('kernels_sha16')}, this run: {sha})"}
                    continue
                if key == "traffic":
                    out["traffic"] = d["kernels"][dom]["hbm_bytes_per_launch_raw"]
                else:
                    out["valu"] = d
                out[key + "_from"] = {"file": "profiles/" + name, "kernels_sha16": sha, "git": d.get("git"),
                                      "box": d.get("box")}
                break
            except Exception:
                continue
    return out


def finish_rank0(args, world, st):
    """correctness gate + CPU baseline (oracle = checker / baseline only), then the ONE JSON line"""
    from nim_groth16_amd import bn128 as F
    from tests.oracle_c import load_oracle
    from tests.parity import check_gpu_proof
    zkey, wits, mask, proofs, ctx = st["zkey"], st["wits"], st["mask"], st["proofs"], st["ctx"]
    shard, inflight = st["shard"], st["inflight"]
    orc = load_oracle()
    cpu_s = []
    for k in range(NWITNESS):
        if proofs[k] is None:
            raise SystemExit(f"FAIL: witness {k} was never proved")
        try:
            cpu_s.append(check_gpu_proof(orc, zkey, wits[k], F.frSeqToMontBytes(wits[k]), mask.r, mask.s, proofs[k], ctx))
        except AssertionError as e:
            raise SystemExit(f"FAIL (witness {k}): {e}")
    log(f"[bench] correctness gate passed for {NWITNESS} distinct witnesses: GPU proof == CPU oracle proof "
        f"(bit-exact), pairing check ok (oracle and GPU verifier)")
    pos = st.get("poseidon")
    if pos is not None:
        try:
            check_gpu_proof(orc, pos["zkey"], pos["wit"], F.frSeqToMontBytes(pos["wit"]), mask.r, mask.s, pos["proof"], ctx)
        except AssertionError as e:
            raise SystemExit(f"FAIL (Poseidon-shaped circuit): {e}")
        log("[bench] Poseidon-shaped circuit: GPU proof == CPU oracle proof (bit-exact), pairing check ok")
    cpu = None
    if not args.no_cpu_baseline and world == 1:     # reported at N = 1 only
        mean_s = sum(cpu_s) / len(cpu_s)
        cpu = {"value": round(1.0 / mean_s, 5), "unit": "proofs/s", "cores": orc.cores(), "kind": "port",
               "sample": f"{len(cpu_s)} full proofs (buildABC + 6 NTT + 4 G1 MSM + 1 G2 MSM), domain "
                         f"2^{args.log2n}, {mean_s:.1f}s each; C restatement, NOT constantine"}
    where = "host (pinned, .wtns layout)" if args.witness == "host" else "HBM"
    out = {
        "metric": "proofs/sec (BN254 Groth16, 2^20-constraint circuit)" if args.log2n == 20 else
                  f"proofs/sec (BN254 Groth16, 2^{args.log2n} domain)",
        "value": round(st["value"], 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(st["dt"] / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong" if shard else "weak", "vs_baseline": None,
        "dtype": "u32 (254-bit Montgomery: 8x32-bit limbs; 9x29-bit limbs in the bucket accumulation)",
        "data": "synthetic",
        "config": {"workload": f"BN254 2^{args.log2n}-constraint synthetic R1CS (squaring chain, m=2^{args.log2n}-2), "
                               "full prove: buildABC + 6 NTT + 4 G1 MSM + 1 G2 MSM, snarkjs flavour, 1 proof/step/GPU"
                               f" ({inflight} proofs in flight per GPU, one shared key)"
                               if not shard else
                               f"BN254 2^{args.log2n}-constraint synthetic R1CS, ONE proof per step, MSMs point-sharded "
                               f"over {world} GPUs + all-gather of partials, quotient {args.quotient}, {inflight} "
                               "sharded proofs in flight",
                   "mode": "shard" if shard else "replica", "nvars": zkey.header.nvars,
                   "domain_log2": args.log2n,
                   "inputs": f"proving key resident in HBM, witness from {where}, rotating over {NWITNESS} "
                             "distinct satisfying witnesses"},
        "roofline": st.get("roof"), "cpu_baseline": cpu,
    }
    extra = st.get("extra", {})
    other = "value_witness_in_hbm" if args.witness == "host" else "value_witness_from_host"
    extra[other] = round(st["value_other"], 4)
    extra["value_runs"] = st["value_runs"]
    extra["value_is"] = f"median of {len(st['value_runs'])} timed regions of {args.steps} proofs each"
    extra["proof_latency_ms_single_in_flight"] = round(st["lat_ms"], 3)
    extra["proofs_in_flight_per_gpu"] = inflight
    extra["keys_resident_per_gpu"] = 1
    extra["kernels_sha16"] = lib_sha16()
    if pos is not None:
        extra["value_poseidon_shape"] = pos["line"]["value"]
        extra["poseidon_shape"] = pos["line"]
    elif st.get("poseidon_error"):
        extra["value_poseidon_shape"] = None
        extra["poseidon_shape"] = {"skipped": st["poseidon_error"]}
    out.update(extra)
    print(json.dumps(out), flush=True)


def dry_run(args, rank, world):
    """rendezvous + barrier + max-over-ranks on gloo, no GPU work: proves that `python bench.py --gpus N` brings up
    N ranks that find each other and that rank 0 alone prints the line"""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        injected_failure(args, rank, None)
    except BaseException as e:       # noqa: BLE001
        fail_fast(e, rank, world)
    if world > 1:
        dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert int(t.item()) == world
    if rank == 0:
        print(json.dumps({"metric": "proofs/sec (dry run: launch plumbing only)", "value": 0.0, "unit": "proofs/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "config": {"mode": args.mode}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def valu_roofline(dom, isolated_ms, clock_ghz, static):
    """The ALU-side roofline of the dominant kernel.  This run contributes the isolated launch duration and the
    shader clock the chip sustained during those launches (g16_profile_clock: s_memtime / s_memrealtime stamped
    by the kernel's own workgroups); the VALU wave-instruction counts of one launch and the issue costs per instruction class
    come from the committed counter pass / micro-benchmarks of this very build (static_inputs drops them otherwise)."""
    inp = static.get("valu") if static else None
    if isolated_ms is None or not inp or dom not in inp.get("kernels", {}):
        return {"bound": "valu-issue", "kernel": dom, "achieved_ms": None if isolated_ms is None else round(isolated_ms, 4),
                "sustained_clock_ghz": round(clock_ghz, 4) if clock_ghz else None, "frac_mix": None,
                "frac_multiply_only": None, "inputs_from": static.get("valu_from") if static else None,
                "note": "no counter pass of this build is committed: instruction counts withheld"}
    k = inp["kernels"][dom]
    simds = 1024
    clock = clock_ghz if clock_ghz and clock_ghz > 0.5 else k["sustained_clock_ghz"]
    wave_insts = k["valu_wave_insts_per_launch"]
    mix_cyc = k["mix_issue_cycles_per_inst"]           # instruction-mix-weighted issue cost, real cycles
    mad_insts = k["mad_u64_wave_insts_per_launch"]
    mad_cyc = inp["issue_cycles"]["v_mad_u64_u32"]
    t_mix = wave_insts / simds * mix_cyc / (clock * 1e9) * 1e3
    t_mul = mad_insts / simds * mad_cyc / (clock * 1e9) * 1e3
    return {"bound": "valu-issue", "kernel": dom, "achieved_ms": round(isolated_ms, 4),
            "valu_wave_insts_per_launch": wave_insts, "mad_u64_wave_insts_per_launch": mad_insts,
            "sustained_clock_ghz": round(clock, 4),
            "clock_source": "s_memtime / s_memrealtime inside this run's isolated launches (g16_profile_clock)" if clock is clock_ghz else
                            "counter pass (GRBM_GUI_ACTIVE / duration)",
            "mix_issue_cycles_per_inst": mix_cyc, "mad_u64_issue_cycles": mad_cyc,
            "bound_ms_mix": round(t_mix, 4), "frac_mix": round(t_mix / isolated_ms, 4),
            "bound_ms_multiply_only": round(t_mul, 4), "frac_multiply_only": round(t_mul / isolated_ms, 4),
            "inputs_from": static.get("valu_from")}


if __name__ == "__main__":
    main()
