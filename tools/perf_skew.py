#!/usr/bin/env python3
"""Worst-case scalar distributions for the bucket scheduler at 2^20 (registered G1 set): every scalar equal (13 buckets
hold everything), 2 / 16 distinct values, and all scalars tiny (< 2^8: a single window).  Checks the result through
known discrete logs and prints the time per MSM."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from nim_groth16_amd import Context
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.synthetic import _fr_stream
    R = F.primeR
    n = 1 << 20
    ctx = Context(0)
    ks = _fr_stream(1, n)
    pts = ctx.fixed_base(1, F.frSeqToMontBytes(ks))
    h = ctx.register_points(1, pts, n)
    big = _fr_stream(9, 16)
    v256, v4096 = _fr_stream(10, 256), _fr_stream(11, 4096)
    cases = {"all equal": [big[0]] * n, "2 values": [big[i & 1] for i in range(n)],
             "16 values": [big[i & 15] for i in range(n)], "all < 2^8": [(i * 7 + 1) & 255 for i in range(n)],
             "256 values": [v256[i & 255] for i in range(n)],      # 3328 buckets of 4096 entries each
             "4096 values": [v4096[i & 4095] for i in range(n)],   # 53 k buckets of 256 entries each
             "uniform": _fr_stream(2, n)}
    for name, sc in cases.items():
        sb = F.frSeqToMontBytes(sc)
        exp = ctx.fixed_base(1, F.frSeqToMontBytes([sum(s * k for s, k in zip(sc, ks)) % R]))
        assert ctx.msm_points(h, sb) == exp, name
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.msm_points(h, sb)
        print(f"{name:12s}: {(time.perf_counter() - t0) / 5 * 1e3:7.2f} ms per 2^20 G1 MSM (host scalars)", flush=True)
    h.release()


if __name__ == "__main__":
    main()
