#!/usr/bin/env python3
"""ALU-side roofline inputs of the hot kernels -> profiles/r02_valu_roofline_inputs.json (read by bench.py).

Three measured ingredients, no assumed clock anywhere:
  1. issue cost of every instruction class in REAL shader cycles: tools/ubench_int (s_memtime deltas)
       ./tools/ubench_int > profiles/r02_ubench_int_issue_rates.txt
  2. the instruction mix of each kernel's hot loop: static count over the disassembly of the built object
     (llvm-objdump of the gfx950 code object inside nim_groth16_amd/csrc/msm_g{1,2}_accum.o, ntt.o); the hot loop is
     the backward branch with the longest span inside the kernel
  3. dynamic VALU wave-instructions per launch, launch duration and sustained clock: one rocprofv3 counter pass
       cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
       rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \\
           SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_r02/valu -o r02 \\
           -- python3 tools/perf.py --only reg --log2n 20 --reps 2
     clock = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back).

  python3 tools/valu_roofline.py gpurun_out/pmc_r02/valu profiles/r02_ubench_int_issue_rates.txt \\
      profiles/r02_valu_roofline_inputs.json

Per kernel the file holds: VALU wave-instructions per launch (PMC), the hot loop's class counts, the mix-weighted
issue cost per instruction, the v_mad_u64_u32 share, duration and clock; bench.py turns them into
  bound_ms_mix           = insts / 1024 SIMDs * mix cycles / clock         (every VALU instruction at its issue cost)
  bound_ms_multiply_only = insts * mad share / 1024 * mad cycles / clock    (nothing but the multiply-adds)."""
import collections
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
SIMDS = 1024

# instruction -> ubench row that prices it (classes of equal issue cost share a row)
PRICE_ROW = [
    (r"^v_mad_u64_u32", "v_mad_u64_u32"), (r"^v_mul_lo_u32", "v_mul_lo_u32"), (r"^v_mul_hi_u32", "v_mul_hi_u32"),
    (r"^v_lshrrev_b64|^v_lshlrev_b64|^v_ashrrev_i64", "v_lshrrev_b64"), (r"^v_lshl_add_u64", "v_lshl_add_u64"),
    (r"^v_add3_u32|^v_or3_b32|^v_and_or_b32|^v_lshl_add_u32|^v_add_lshl_u32|^v_xad_u32|^v_bfe_u32|^v_bfi_b32|^v_perm_b32"
     r"|^v_lshl_or_b32|^v_mad_u32_u24|^v_cndmask_b32_e64|^v_cmp_.*_e64|^v_add_co_u32_e64|^v_addc_co_u32_e64"
     r"|^v_sub_co_u32_e64|^v_subb_co_u32_e64|^v_readlane|^v_writelane|^v_mov_b64", "v_add3_u32"),
    (r"^v_alignbit_b32", "v_alignbit_b32"),
    (r"^v_addc_co_u32|^v_add_co_u32|^v_subb_co_u32|^v_sub_co_u32|^v_subrev_co_u32|^v_subbrev_co_u32", "add_co+addc"),
    (r"^v_", "v_and_b32"),       # plain VOP1/VOP2 (and, add, sub, shifts by a register/immediate, mov, cndmask, cmp)
]


def ubench_table(path):
    """cycles per wave-instruction per SIMD at 8 waves/SIMD, from the ubench text"""
    tab, section = {}, None
    for ln in open(path):
        if ln.startswith("----"):
            section = ln.strip("- \n")
            continue
        m = re.match(r"^(\S.*?)\s+blocks=\s*\d+.*=>\s*([0-9.]+) cycles/winstr/SIMD", ln)
        if m and section and section.startswith("8 waves"):
            tab[m.group(1).strip()] = float(m.group(2))
    return tab


def code_object(obj):
    """extract the gfx950 code object bundled in a hipcc .o and return its disassembly"""
    tmp = f"/tmp/valu_roofline_{os.path.basename(obj)}"
    os.makedirs(tmp, exist_ok=True)
    local = os.path.join(tmp, os.path.basename(obj))
    subprocess.check_call(["cp", obj, local])
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True, cwd=tmp)
    co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(tmp, co)], check=True, capture_output=True,
                          text=True).stdout


def hot_loop_mix(disasm, kernel_regex, whole_kernel=False):
    """static instruction counts of the longest backward-branch loop inside the kernel whose symbol matches
    (whole_kernel: of the entire kernel -- for kernels made of several comparable loops, like the NTT passes)"""
    lines = disasm.split("\n")
    start = end = None
    for i, ln in enumerate(lines):
        if re.match(r"^[0-9a-f]+ <", ln):
            if start is not None and end is None:
                end = i
            if start is None and re.search(kernel_regex, ln):
                start = i
    ins = []
    for ln in lines[start:end]:
        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):", ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if whole_kernel:
        return collections.Counter(op for _, op, _ in ins)
    best = (0, 0, 0)
    for addr, op, args in ins:
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(args.split()[-1])
            if off >= 32768:
                off -= 65536
            tgt = addr + 4 + 4 * off
            # (a backward branch to the kernel's exit block is not a loop: its range would hold the s_endpgm)
            if tgt < addr and addr - tgt > best[0] and not any(o == "s_endpgm" and tgt <= a <= addr for a, o, _ in ins):
                best = (addr - tgt, tgt, addr)
    _, lo, hi = best
    counts = collections.Counter(op for addr, op, _ in ins if lo <= addr <= hi)
    return counts


def price(counts, tab):
    """-> (VALU instructions, mix-weighted cycles per VALU instruction, v_mad_u64_u32 share, per-class breakdown)"""
    cyc = n = mad = 0
    classes = collections.Counter()
    for op, c in counts.items():
        if not op.startswith("v_"):
            continue
        row = next(r for pat, r in PRICE_ROW if re.match(pat, op))
        per = tab[row] / 2 if row == "add_co+addc" and False else tab[row]
        cyc += c * per
        n += c
        classes[row] += c
        if op.startswith("v_mad_u64_u32"):
            mad += c
    return n, cyc / n, mad / n, dict(classes)


def short(name):
    base = re.split(r"[<(]", name.replace("void ", "").replace("g16::", "").replace("(anonymous namespace)::", ""))[0].strip()
    if base.startswith("msm_") and ("Curve" in name or "Fp2" in name or "Field" in name):
        base += "_g2" if "Fp2" in name else "_g1"
    return base


def pmc(root, tag="r02"):
    rows = list(csv.DictReader(open(os.path.join(root, f"{tag}_counter_collection.csv"))))
    kt = list(csv.DictReader(open(os.path.join(root, f"{tag}_kernel_trace.csv"))))
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt}
    per, names = collections.defaultdict(dict), {}
    for r in rows:
        d = r["Dispatch_Id"]
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0) + float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
    agg = collections.defaultdict(list)
    for d, c in per.items():
        if d in dur:
            agg[short(names[d])].append((dur[d], c))
    out = {}
    for k, lst in agg.items():
        lst.sort(key=lambda t: t[0])
        med = lst[len(lst) // 2][0]
        lst = [t for t in lst if t[0] >= 0.5 * med]      # drop toy-size launches (the known-answer MSM / NTT of g16_selftest)
        lst = lst[:max(1, len(lst) * 3 // 4)]            # drop the slowest quarter (first-touch / cold launches)
        n = len(lst)
        us = sum(t[0] for t in lst) / n / 1e3
        c = {key: sum(t[1].get(key, 0) for t in lst) / n for key in lst[0][1]}
        cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
        out[k] = {"launches": n, "duration_us": round(us, 1), "sustained_clock_ghz": round(cyc / us / 1e3, 3) if us else None,
                  "valu_wave_insts_per_launch": int(c.get("SQ_INSTS_VALU", 0)), "waves": int(c.get("SQ_WAVES", 0)),
                  "cycles_per_valu_inst_per_simd": round(cyc * SIMDS / c["SQ_INSTS_VALU"], 3) if c.get("SQ_INSTS_VALU") else None,
                  "counters": {key: int(v) for key, v in c.items()}}
    return out


def main():
    pmc_root, ubench_path, out_path = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else "r02"          # the -o prefix given to rocprofv3
    tab = ubench_table(ubench_path)
    counters = pmc(pmc_root, tag)
    for extra in sys.argv[5:]:                                 # further counter passes (e.g. .../valu_abc: buildABC)
        for k, v in pmc(extra, tag).items():
            counters.setdefault(k, v)
    csrc = os.path.join(ROOT, "nim_groth16_amd", "csrc")
    loops = {"msm_accum_g1": ("msm_g1_accum.o", r"msm_accum"), "msm_accum_g2": ("msm_g2_accum.o", r"msm_accum"),
             "ntt_pass": ("ntt.o", r"ntt_passILi"), "ntt_last_pass_abc": ("ntt.o", r"ntt_last_pass_abc"),
             "spmv_binned": ("spmv.o", r"spmv_binnedILi2ELb1")}
    kernels = {}
    for k, (obj, rx) in loops.items():
        if k not in counters:
            continue
        counts = hot_loop_mix(code_object(os.path.join(csrc, obj)), rx, whole_kernel=k.startswith("ntt") or k.startswith("spmv"))
        n, mix, mad_share, classes = price(counts, tab)
        c = counters[k]
        kernels[k] = {**{kk: vv for kk, vv in c.items() if kk != "counters"},
                      "hot_loop_valu_insts": n, "hot_loop_s_nop": counts.get("s_nop", 0),
                      "hot_loop_classes": classes, "mix_issue_cycles_per_inst": round(mix, 3),
                      "mad_u64_share_of_valu": round(mad_share, 4),
                      "mad_u64_wave_insts_per_launch": int(c["valu_wave_insts_per_launch"] * mad_share),
                      "counters": c["counters"]}
        t_mix = c["valu_wave_insts_per_launch"] / SIMDS * mix / (c["sustained_clock_ghz"] * 1e9) * 1e6
        t_mul = kernels[k]["mad_u64_wave_insts_per_launch"] / SIMDS * tab["v_mad_u64_u32"] / (c["sustained_clock_ghz"] * 1e9) * 1e6
        kernels[k]["bound_us_mix"], kernels[k]["frac_mix"] = round(t_mix, 1), round(t_mix / c["duration_us"], 4)
        kernels[k]["bound_us_multiply_only"], kernels[k]["frac_multiply_only"] = round(t_mul, 1), round(t_mul / c["duration_us"], 4)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pmc_traffic import provenance
    doc = {**provenance(),
           "how": "tools/valu_roofline.py (recipe in its docstring): ubench issue costs in real cycles (s_memtime), static "
                  "hot-loop mix from the built objects, dynamic VALU wave-instructions / duration / clock from one rocprofv3 "
                  "--pmc pass of tools/perf.py --only reg --log2n 20 (registered G1 / G2 MSM, NTT, quotient; each kernel alone on the GPU)",
           "issue_cycles": tab, "simds": SIMDS, "kernels": kernels,
           "other_kernels": {k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in counters.items()
                             if k not in kernels}}
    json.dump(doc, open(out_path, "w"), indent=1)
    for k, v in kernels.items():
        print(k, {kk: v[kk] for kk in ("duration_us", "sustained_clock_ghz", "valu_wave_insts_per_launch",
                                       "cycles_per_valu_inst_per_simd", "mix_issue_cycles_per_inst", "frac_mix",
                                       "mad_u64_share_of_valu", "frac_multiply_only")})


if __name__ == "__main__":
    main()
