import csv, collections, json, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(f"{d}/r01_counter_collection.csv")))
kt = list(csv.DictReader(open(f"{d}/r01_kernel_trace.csv")))
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt}
per = collections.defaultdict(dict); names = {}
for r in rows:
    per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    names[r["Dispatch_Id"]] = r["Kernel_Name"]
for dd, c in per.items():
    k = names[dd]
    if "msm_accum" not in k: continue
    g = "g2" if "Fp2" in k else "g1"
    cyc = c["GRBM_GUI_ACTIVE"] / 8; us = dur[dd] / 1000
    print(g, "dur_us", round(us,1), "clk", round(cyc/us/1000,3), "valu", int(c["SQ_INSTS_VALU"]), "cyc/inst/simd", round(cyc*1024/c["SQ_INSTS_VALU"],2), "waves", int(c["SQ_WAVES"]), {k2:int(v) for k2,v in c.items() if k2 not in ("GRBM_GUI_ACTIVE","SQ_INSTS_VALU","SQ_WAVES")})
