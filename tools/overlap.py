"""How well a multi-proof run keeps the issue-bound accumulate kernels fed: from a rocprofv3 --kernel-trace CSV
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- /tmp/g16ab/ab_prove -l <lib> -z ... -w ... -k 48 -r 1
  python3 tools/overlap.py gpurun_out/tl/tl_kernel_trace.csv
prints, over the middle of the run: the share of wall time with at least one accumulate kernel resident, the mean
number of resident kernels, and per kernel class the launches, the mean in-run duration and the summed duration per
proof (compare with the stand-alone times of tools/perf.py: the ratio is the time a kernel spends waiting for slots)."""
import collections
import csv
import re
import sys


def short(name):
    base = re.split(r"[<(]", name.replace("void ", "").replace("g16::", "").replace("(anonymous namespace)::", ""))[0].strip()
    if base.startswith("msm_") and ("Curve" in name or "Fp2" in name):
        base += "_g2" if "Fp2" in name else "_g1"
    return base


def main(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    lo, hi = t0 + (t1 - t0) * 0.35, t0 + (t1 - t0) * 0.9
    win = [e for e in ev if e[0] >= lo and e[1] <= hi]
    nproofs = sum(1 for e in win if e[2] == "prove_combine_kernel")
    T = hi - lo
    pts = []
    for s, e, n in win:
        a = 1 if n.startswith("msm_accum") else 0
        pts.append((s, 1, a))
        pts.append((e, -1, -a))
    pts.sort()
    act = acc = 0
    last = lo
    busy = with_acc = conc = 0.0
    for t, d, a in pts:
        dt = t - last
        if act > 0:
            busy += dt
        if acc > 0:
            with_acc += dt
        conc += dt * act
        act += d
        acc += a
        last = t
    print(f"window {T / 1e6:.1f} ms, {nproofs} proofs ({T / 1e6 / max(nproofs, 1):.3f} ms per proof), {len(win)} launches "
          f"({len(win) / max(nproofs, 1):.0f} per proof)")
    print(f"some kernel resident {100 * busy / T:.1f} % of the time; an accumulate kernel resident {100 * with_acc / T:.1f} %; "
          f"mean resident kernels {conc / T:.2f}")
    by = collections.defaultdict(list)
    for s, e, n in win:
        by[n].append(e - s)
    print(f"{'kernel':28s} {'launches/proof':>14s} {'mean ms in run':>15s} {'sum ms/proof':>13s}")
    for n, l in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n:28s} {len(l) / max(nproofs, 1):14.1f} {sum(l) / len(l) / 1e6:15.3f} {sum(l) / 1e6 / max(nproofs, 1):13.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
