#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of tools/pmc_shapes.py: per (kernel, grid) the mean of every counter over the launches.
usage: python tools/pmc_sq_summary.py OUT.json counter_collection.csv [more.csv ...]
Derived (MI355X_MICROARCH.md, "rocprofv3 PMC slots"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES-equivalent) is
reported as MFMA-busy cycles per CU-active cycle when both counters are present."""
import csv, json, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r"gemm_f16_kernelILi(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d)ELi(\d)E", name) or \
        re.search(r"gemm_f16_kernel<(\d+), (\d+), \d+, \d+, (\d+), (\d+)>", name)          # mangled / demangled spelling
    if m:
        return f"gemm<{m.group(1)}x{m.group(2)},mode{m.group(3)},ns{m.group(4)}>"
    for k in ("attention_kernel", "temporal_attention", "gn_apply", "gn_partial", "layernorm"):
        if k in name:
            return k
    return None


# the GEMM jobs of tools/pmc_shapes.py in launch order, REPS launches each: persistent workgroups give every big launch the same grid,
# so a shape is identified by its position in the dispatch sequence of the pass
import ast, os
_src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_shapes.py")).read()
JOB_LABELS = ast.literal_eval(re.search(r"JOB_LABELS = (\[.*?\])", _src, flags=re.S).group(1))
REPS = int(re.search(r"^REPS = (\d+)", _src, flags=re.M).group(1))

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
dur = defaultdict(lambda: [0, 0.0])
for path in sys.argv[2:]:
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    gemm_ids = sorted({int(r["Dispatch_Id"]) for r in rows if (short(r["Kernel_Name"]) or "").startswith("gemm<")})
    job_of = {did: i // REPS for i, did in enumerate(gemm_ids)}
    for row in rows:
        k = short(row["Kernel_Name"])
        if k is None:
            continue
        if k.startswith("gemm<"):
            j = job_of[int(row["Dispatch_Id"])]
            key = f"{k} {JOB_LABELS[j] if j < len(JOB_LABELS) else 'job%d' % j}"
        else:
            key = f"{k} grid={row['Grid_Size']}"
        a = acc[key][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
        dd = dur[key]
        dd[0] += 1
        dd[1] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
out = {}
for key, ctrs in acc.items():
    e = {c: v[1] / v[0] for c, v in ctrs.items()}
    e["launches_seen"] = max(v[0] for v in ctrs.values())
    e["avg_us_profiled"] = dur[key][1] / dur[key][0] / 1e3
    w = e.get("SQ_WAVE_CYCLES")
    if w:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM"):
            if c in e:
                e[c + "_frac_of_wave_cycles"] = e[c] / w
    if "SQ_INSTS_VALU" in e and "SQ_INSTS_MFMA" in e and e["SQ_INSTS_MFMA"]:
        e["valu_per_mfma"] = (e["SQ_INSTS_VALU"] - e["SQ_INSTS_MFMA"]) / e["SQ_INSTS_MFMA"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CU_CYCLES" in e and e["SQ_BUSY_CU_CYCLES"]:
        e["mfma_busy_per_cu_cycle"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * e["SQ_BUSY_CU_CYCLES"])
    if "SQ_LDS_BANK_CONFLICT" in e and "SQ_LDS_IDX_ACTIVE" in e and e["SQ_LDS_IDX_ACTIVE"]:
        e["lds_bank_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
    if "GRBM_GUI_ACTIVE" in e:
        e["effective_clock_ghz"] = e["GRBM_GUI_ACTIVE"] / 8.0 / (e["avg_us_profiled"] * 1e3)
    out[key] = e
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
