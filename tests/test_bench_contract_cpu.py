"""The committed bench line (profiles/) carries every field the driver's contract names, incl. the roofline and
cpu_baseline objects -- a cheap guard against accidentally dropping one when bench.py is edited."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_2p20_final.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "proofs/s" and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong")
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"] * d["n_gpus"]) < 0.01 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    assert "witness from host" in d["config"]["inputs"] and d["keys_resident_per_gpu"] == 1
    # round 3: the fraction comes from the UNCONTENDED launch, the contended figure is a named extra, and every
    # counter-derived number names the build it was measured on -- the build of this very run
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert r["contended_launch_ms_in_timed_region"] > r["avg_launch_ms"]
    assert r["inputs_from"]["kernels_sha16"] == d["kernels_sha16"] == d["roofline_valu"]["inputs_from"]["kernels_sha16"]
    v = d["roofline_valu"]
    assert "g16_profile_clock" in v["clock_source"] and 1.5 < v["sustained_clock_ghz"] < 2.6 and 0.8 < v["frac_mix"] < 1.0
    # round 4: the median of repeated timed regions, the share of a step that is bucket accumulation, the gather roof
    assert len(d["value_runs"]) >= 5 and sorted(d["value_runs"])[len(d["value_runs"]) // 2] == d["value"]
    assert 0.5 < d["overlap_efficiency"] < 1.0 and abs(d["overlap_efficiency"] - d["accum_ms_per_proof"] / d["ms_per_step"]) < 1e-3
    g = d["roofline_gather"]
    assert g["hbm_bytes_per_launch_by_counters"] == r["traffic"] and 0.2 < g["frac"] < 1.0
    # round 5: BASELINE config 5's workload shape as an extra key; `value` is the squaring chain's, untouched
    p = d["poseidon_shape"]
    assert d["value_poseidon_shape"] == p["value"] == sorted(p["runs"])[1] and p["unit"] == "proofs/s"
    assert p["ncoeffs_per_domain_row"] > 10 and p["coefficient_dictionary_values"] > 0 and "NOT circomlib" in p["circuit"]
    assert sum(p["rows_by_terms"].values()) == 2 << d["config"]["domain_log2"] and p["rows_by_terms"]["L<=32"] > 0
    assert p["points_at_infinity"]["B1"] == p["points_at_infinity"]["B2"] > 0.3 * p["nvars"]
    assert "squaring chain" in d["config"]["workload"]


def test_valu_roofline_inputs_are_consistent():
    """profiles/r05_valu_roofline_inputs.json (tools/valu_roofline.py) -> bench.py's roofline_valu object"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    inp = json.load(open(os.path.join(ROOT, "profiles", "r05_valu_roofline_inputs.json")))
    assert inp["kernels_sha16"] and inp["git"]
    # the committed counters belong to the library in the tree (when it is built): a kernel edit must come with a new
    # counter pass (tools/profile_session.sh), or bench.py will rightly report nulls
    from nim_groth16_amd._lib import device_code_sha16, lib_path
    if os.path.exists(lib_path()):
        assert inp["kernels_sha16"] == device_code_sha16(), \
            "profiles/r05_valu_roofline_inputs.json was measured on another build: re-run tools/profile_session.sh pmc + " \
            "tools/valu_roofline.py / tools/pmc_traffic.py (bench.py reports nulls for counter-derived inputs until then)"
        hbm = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_hbm_traffic_2p20.json")))
        assert hbm["kernels_sha16"] == device_code_sha16()
    k = inp["kernels"]["msm_accum_g1"]
    assert 0.5 < k["mad_u64_share_of_valu"] < 0.7 and 3.5 < k["mix_issue_cycles_per_inst"] < 4.5
    assert 4.0 < inp["issue_cycles"]["v_mad_u64_u32"] < 5.0 and 1.8 < k["sustained_clock_ghz"] < 2.5
    r = bench.valu_roofline("msm_accum_g1", k["duration_us"] / 1e3, k["sustained_clock_ghz"],
                            {"valu": inp, "valu_from": {"file": "profiles/r05_valu_roofline_inputs.json"}})
    for key in ("bound", "achieved_ms", "valu_wave_insts_per_launch", "sustained_clock_ghz", "bound_ms_mix", "frac_mix",
                "bound_ms_multiply_only", "frac_multiply_only"):
        assert key in r, key
    assert abs(r["frac_mix"] - k["frac_mix"]) < 0.01 and r["frac_multiply_only"] < r["frac_mix"] < 1.0
    # inputs measured on another build are withheld, not silently reused
    r = bench.valu_roofline("msm_accum_g1", 0.9, 2.0, {"valu": None, "valu_from": {"dropped": "other build"}})
    assert r["frac_mix"] is None and r["sustained_clock_ghz"] == 2.0


def test_bench_source_emits_the_same_keys():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"',
              '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"', '"traffic"',
              '"frac"', '"cores"', '"kind"', '"sample"', '"roofline_valu"', '"frac_mix"', '"frac_multiply_only"',
              '"roofline_gather"', '"value_runs"', '"accum_ms_per_proof"', '"overlap_efficiency"',
              '"value_poseidon_shape"', '"poseidon_shape"'):
        assert k in src, k


def test_overlap_efficiency_is_per_gpu_for_any_world_size():
    """in replica mode EVERY rank proves `steps` proofs in the timed region, so a GPU's time per proof is
    region / steps whatever the world size -- the value must not shrink by the number of GPUs"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    region_s, steps, acc_ms = 2.4, 288, 6.5
    one = bench.overlap_efficiency(acc_ms, region_s, steps)
    assert abs(one - acc_ms / (region_s / steps * 1e3)) < 1e-4 and 0.7 < one < 0.8
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def overlap_efficiency"):src.index("def poseidon_shape")]
    assert "world" not in body.split('"""')[2]                    # the formula has no world factor
    # an 8-GPU replica line: value = 8 x per-GPU rate, ms_per_step = region / steps, same efficiency as one GPU
    value8, ms_per_step8 = 8 * steps / region_s, region_s / steps * 1e3
    assert abs(bench.overlap_efficiency(acc_ms, region_s, steps) - acc_ms / ms_per_step8) < 1e-4
    assert abs(value8 - 8e3 / ms_per_step8) < 1e-6


def test_accumulate_hot_loops_keep_their_instruction_budget():
    """Static regression guard on the built objects (tools/valu_roofline.py: llvm-objdump of the gfx950 code object):
    one mixed addition of msm_accum<G1> is 6 mul + 2 sqr + one 2-term dot product in the 9x29 field = 1467
    v_mad_u64_u32 (6*162 + 2*126 + 243), and the whole loop body stays within the budget the roofline accounting
     is written for.  Skipped when the library has not been built."""
    import sys
    import pytest
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import valu_roofline as V
    csrc = os.path.join(ROOT, "nim_groth16_amd", "csrc")
    if not os.path.exists(os.path.join(csrc, "msm_g1_accum.o")) or not os.path.exists(os.path.join(V.LLVM, "llvm-objdump")):
        pytest.skip("objects or llvm-objdump not present")
    g1 = V.hot_loop_mix(V.code_object(os.path.join(csrc, "msm_g1_accum.o")), r"msm_accum")
    valu = sum(c for op, c in g1.items() if op.startswith("v_"))
    assert g1["v_mad_u64_u32"] == 6 * 162 + 2 * 126 + 243 == 1467
    assert valu <= 2500, valu                                # 2431 when this was written
    assert g1.get("s_nop", 0) <= 300, g1.get("s_nop")        # 249: column chains as asm statements (1261 with per-product pins)
    assert not any(op.startswith("scratch_store") for op in g1), "the G1 accumulate loop spills"
    g2 = V.hot_loop_mix(V.code_object(os.path.join(csrc, "msm_g2_accum.o")), r"msm_accum")
    assert 4300 <= g2["v_mad_u64_u32"] <= 4400 and sum(c for op, c in g2.items() if op.startswith("v_")) <= 6700
    assert g2.get("s_nop", 0) <= 300, g2.get("s_nop")        # 280: chain pairs as asm statements (956 with pinned C++ pairs)
    assert not any(op.startswith("scratch_store") for op in g2), "the G2 accumulate loop spills"
