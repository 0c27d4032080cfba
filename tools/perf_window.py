"""accumulate / heavy / reduce times of one registered G1 MSM (2^20) for several table window sizes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nim_groth16_amd import Context
from tests.oracle_c import load_oracle
from tools.perf import rand_fr_mont_bytes

n = 1 << 20
orc = load_oracle()
p1 = orc.fixed_base(1, rand_fr_mont_bytes(n, 1))
sb = rand_fr_mont_bytes(n, 2)
want = None
for c in (16, 17, 18, 19, 20, 21):
    os.environ["G16_TABLE_WINDOW"] = str(c)
    ctx = Context(0)
    d_s = torch.frombuffer(bytearray(sb), dtype=torch.uint8).cuda()
    d_p = torch.frombuffer(bytearray(p1), dtype=torch.uint8).cuda()
    h = ctx.register_points(1, d_p.data_ptr(), n, device=True)
    r = ctx.msm_points(h, d_s.data_ptr(), device=True)
    want = want or r
    assert r == want
    ctx.profile(True); ctx.profile_reset()
    for _ in range(3):
        ctx.msm_points(h, d_s.data_ptr(), device=True)
    rep = ctx.profile_report(); ctx.profile(False)
    tot = sum(v["total_ms"] for v in rep.values()) / 3
    g = lambda k: rep.get(k, {"total_ms": 0})["total_ms"] / 3
    print(f"c={c} W={254//c+1} total {tot:.3f} ms | accum {g('msm_accum_g1'):.3f} heavy {g('msm_heavy_g1'):.3f} "
          f"reduce1 {g('msm_reduce1_g1'):.3f} reduce2 {g('msm_reduce2_g1'):.3f} fold {g('msm_fold_g1'):.3f} "
          f"sort {g('msm_part_count')+g('msm_part_scatter')+g('msm_bucket_sort'):.3f}", flush=True)
    h.release(); ctx.close()
