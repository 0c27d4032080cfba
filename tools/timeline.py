"""Per-stream timeline of the single-in-flight proofs of a bench.py run traced with rocprofv3 --kernel-trace:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline
  python3 tools/timeline.py gpurun_out/tl/tl_kernel_trace.csv
Prints, for the three latency-measurement proofs, when each stream starts / ends and its long kernels (ms from the
proof's first GPU activity).
  python3 tools/timeline.py <trace.csv> --gap [min_us]
splits the trace at idle gaps instead (a run without combine kernels, e.g. tools/perf_shard.py: one window per
synchronised repetition) and prints every kernel of the last window longer than min_us (default 20)."""
import collections
import csv
import re
import sys


def short(name):
    base = re.split(r"[<(]", name.replace("void ", "").replace("g16::", ""))[0].strip()
    if base.startswith("msm_") and "Fp2" in name:
        base += "_g2"
    elif base.startswith("msm_") and "Curve" in name:
        base += "_g1"
    return base


rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Stream_Id"]) for r in rows)
if "--gap" in sys.argv:
    k = sys.argv.index("--gap")
    min_ns = (float(sys.argv[k + 1]) if len(sys.argv) > k + 1 else 20.0) * 1e3
    wins, cur, hi = [], [], 0
    for e in ev:
        if cur and e[0] > hi + 150_000:      # the host's synchronise + relaunch leaves the GPU idle for > 0.15 ms
            wins.append(cur)
            cur = []
        cur.append(e)
        hi = max(hi, e[1])
    wins.append(cur)
    win = wins[-1]
    t0, end = win[0][0], max(e[1] for e in win)
    print(f"---- last window of {len(wins)}: {len(win)} kernels, span {(end - t0) / 1e6:.2f} ms")
    by = collections.defaultdict(list)
    for s, e, n, st in win:
        by[st].append((s, e, n))
    for st, l in sorted(by.items(), key=lambda kv: kv[1][0][0]):
        print(f" stream {st}: {len(l)} kernels, {(l[0][0] - t0) / 1e6:.2f} -> {(l[-1][1] - t0) / 1e6:.2f} ms, busy "
              f"{sum(e - s for s, e, n in l) / 1e6:.2f}")
        for s, e, n in l:
            if e - s > min_ns:
                print(f"      {n:22s} {(s - t0) / 1e6:6.3f} -> {(e - t0) / 1e6:6.3f} ({(e - s) / 1e6:.3f})")
    sys.exit(0)
comb = [i for i, e in enumerate(ev) if e[2] == "prove_combine_kernel"]
for ci in comb[-3:]:        # bench.py's last three proofs: its single-in-flight latency measurement
    end = ev[ci][1]
    prev = [e for e in ev if e[2] == "prove_combine_kernel" and e[1] < ev[ci][0]][-1][1]
    win = [e for e in ev if e[0] >= prev and e[1] <= end]
    t0 = win[0][0]
    print(f"---- proof: span {(end - t0) / 1e6:.2f} ms (gap to the previous proof's end {(t0 - prev) / 1e6:.2f} ms)")
    by = collections.defaultdict(list)
    for s, e, n, st in win:
        by[st].append((s, e, n))
    for st, l in sorted(by.items(), key=lambda kv: kv[1][0][0]):
        print(f" stream {st}: {len(l)} kernels, {(l[0][0] - t0) / 1e6:.2f} -> {(l[-1][1] - t0) / 1e6:.2f} ms, busy "
              f"{sum(e - s for s, e, n in l) / 1e6:.2f}; first {l[0][2]}, last {l[-1][2]}")
        for s, e, n in l:
            if e - s > 2e5:
                print(f"      {n:22s} {(s - t0) / 1e6:6.2f} -> {(e - t0) / 1e6:6.2f} ({(e - s) / 1e6:.2f})")
