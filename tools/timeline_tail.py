"""Every kernel of the last `ms` milliseconds of a rocprofv3 --kernel-trace CSV, per stream (ms from the first one):
  python3 tools/timeline_tail.py trace.csv [ms=3.5]
For runs whose repetitions follow each other without idle gaps (tools/perf_shard.py)."""
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
span = float(sys.argv[2]) if len(sys.argv) > 2 else 3.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Stream_Id"]) for r in rows)
end = max(e[1] for e in ev)
win = [e for e in ev if e[0] > end - span * 1e6]
t0 = win[0][0]
by = collections.defaultdict(list)
for s, e, n, st in win:
    by[st].append((s, e, n))
for st, l in sorted(by.items(), key=lambda kv: kv[1][0][0]):
    print(f"stream {st}: {len(l)} kernels, busy {sum(e - s for s, e, n in l) / 1e6:.2f} ms")
    for s, e, n in l:
        print(f"   {n:26s} {(s - t0) / 1e6:6.3f} -> {(e - t0) / 1e6:6.3f} ({(e - s) / 1e3:.0f} us)")
