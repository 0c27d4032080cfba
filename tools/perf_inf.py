#!/usr/bin/env python3
"""Registered G1 / G2 MSM at 2^20 against the fraction of (0,0) points in the set (real snarkjs keys hold the point at
infinity for every wire absent from a matrix).  Run once as is and once with G16_INF_COMPACT=101 (compaction off:
every (0,0) entry costs a loop trip, rounds 1-2) for the before/after rows; results are checked against each other
through the closed form (known discrete logs)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import numpy as np
    import torch
    from nim_groth16_amd import Context
    from nim_groth16_amd import bn128 as F
    from nim_groth16_amd.synthetic import _fr_stream
    R = F.primeR
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << log2n
    ctx = Context(0)
    ks, sc = _fr_stream(1, n), _fr_stream(2, n)
    sb = torch.frombuffer(bytearray(F.frSeqToMontBytes(sc)), dtype=torch.uint8).cuda()
    print(f"G16_INF_COMPACT={os.environ.get('G16_INF_COMPACT', '(default 10)')}", flush=True)
    for group in (1, 2):
        psz = 64 * group
        base = np.frombuffer(ctx.fixed_base(group, F.frSeqToMontBytes(ks)), dtype=np.uint8).reshape(n, psz)
        order = np.random.default_rng(7).permutation(n)
        for frac in (0.0, 0.25, 0.5, 0.75, 0.9):
            dead = order[: int(frac * n)]
            pts = base.copy()
            pts[dead] = 0
            h = ctx.register_points(group, pts.tobytes(), n)
            deadset = np.zeros(n, dtype=bool)
            deadset[dead] = True
            e = sum(s * k for s, k, d in zip(sc, ks, deadset) if not d) % R
            assert ctx.msm_points(h, sb.data_ptr(), device=True) == ctx.fixed_base(group, F.frSeqToMontBytes([e])), frac
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                ctx.msm_points(h, sb.data_ptr(), device=True)
            dt = (time.perf_counter() - t0) / 5 * 1e3
            print(f"G{group} 2^{log2n} registered MSM, {100 * frac:3.0f} % (0,0) points ({h.inf_count()} counted): "
                  f"{dt:6.3f} ms", flush=True)
            h.release()


if __name__ == "__main__":
    main()
