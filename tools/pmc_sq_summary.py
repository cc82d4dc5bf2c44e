#!/usr/bin/env python3
"""Per-kernel means of the shader-side PMC passes written by tools/pmc_sq.sh (one csv per counter group)."""
import csv, json, re, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:-1]:
    per_dispatch = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "rf::" not in name:
            continue
        m = re.search(r"(\w+_kernel)(<[^,>]*,\s*\d+)?", name)
        short = (m.group(1) + (m.group(2) or "")) if m else name[:40]
        per_dispatch[(r["Dispatch_Id"], short)][r["Counter_Name"]] = per_dispatch[(r["Dispatch_Id"], short)].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (_, short), counters in per_dispatch.items():
        for c, v in counters.items():
            out[short][c].append(v)
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()}
for k, cs in res.items():
    wc = cs.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY",
                  "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INST_CYCLES_VMEM"):
            if c in cs:
                cs[c + "_per_wave_cycle"] = round(cs[c] / wc, 4)
    if cs.get("SQ_LDS_IDX_ACTIVE"):
        cs["lds_bank_conflict_frac"] = round(cs.get("SQ_LDS_BANK_CONFLICT", 0.0) / cs["SQ_LDS_IDX_ACTIVE"], 4)
    if cs.get("SQ_WAVES"):
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU"):
            if c in cs:
                cs[c + "_per_wave"] = round(cs[c] / cs["SQ_WAVES"], 1)
json.dump({"note": "means per launch over the dispatches of one bench run (cfg3); counters summed over XCDs/SEs as rocprofv3 reports them",
           "kernels": res}, open(sys.argv[-1], "w"), indent=1)
for k, cs in res.items():
    print(k, {c: v for c, v in cs.items() if c.endswith(("per_wave_cycle", "per_wave", "frac"))})
