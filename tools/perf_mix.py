"""Do the G1 and the G2 bucket accumulations slow each other down when they share the GPU?  Three host threads (one
context each, like three proofs in flight) run registered 2^20 MSMs: (A) all G1, (B) all G2, (C) every thread cycles
G1 G1 G1 G1 G2 -- a proof's mix.  If the kernels only competed for issue slots, a cycle of (C) would take
4 T_G1 + T_G2 of (A) and (B); whatever it takes beyond that is interference (instruction cache: the G2 accumulate loop
is 54 KB, the G1 loop 21 KB, the cache 64 KB per CU pair; register / LDS fragmentation; clocks)."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nim_groth16_amd import Context  # noqa: E402
from tests.oracle_c import load_oracle  # noqa: E402
from tools.perf import rand_fr_mont_bytes  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
orc = load_oracle()
ctxs = [Context(0) for _ in range(3)]
kb = rand_fr_mont_bytes(n, 1)
d_s = torch.frombuffer(bytearray(rand_fr_mont_bytes(n, 2)), dtype=torch.uint8).cuda()
h1 = ctxs[0].register_points(1, orc.fixed_base(1, kb), n)
h2 = ctxs[0].register_points(2, orc.fixed_base(2, kb), n)
torch.cuda.synchronize()


def run(pattern, cycles):
    def work(j):
        torch.cuda.set_device(0)
        for _ in range(cycles):
            for g in pattern:
                ctxs[j].msm_points(h1 if g == 1 else h2, d_s.data_ptr(), device=True)
    th = [threading.Thread(target=work, args=(j,)) for j in range(3)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return (time.perf_counter() - t0) / cycles * 1e3 / 3     # ms per pattern, per thread-equivalent


for name, pat, cyc in (("warm", (1, 2), 3), ("A: G1 only", (1,), 60), ("B: G2 only", (2,), 24), ("C: G1 G1 G1 G1 G2", (1, 1, 1, 1, 2), 16),
                       ("A again", (1,), 60), ("C again", (1, 1, 1, 1, 2), 16)):
    t = run(pat, cyc)
    print(f"{name:22s} {t:8.3f} ms per pattern (3 in flight)", flush=True)
