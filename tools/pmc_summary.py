"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel family.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads, so the read
side is doubled; WRITE_SIZE is exact for 16 B/lane stores.  Infinity-Cache hits are counted, not excluded.
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import csv
import os
import json
import re
import sys
from collections import defaultdict


def family(name):
    m = re.search(r"(gemm_f16_kernel)ILi(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d)E", name) or \
        re.search(r"(gemm_f16_kernel)<(\d+), (\d+), \d+, \d+, (\d+), \d+>", name)      # mangled / demangled spelling
    if m:
        return f"gemm_f16_kernel<{m.group(2)},{m.group(3)},mode{m.group(4)}>"
    for k in ("attention_kernel", "temporal_attention_kernel", "gn_apply_kernel", "gn_partial_kernel", "layernorm_kernel",
              "concat_kernel", "ring_gather_renoise_kernel", "cfg_ddim_scatter_kernel", "ring_gather_kernel", "ring_scatter3_kernel", "renoise_mix_kernel",
              "cfg_ddim_kernel"):
        if k in name:
            return k
    return None


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0, 0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        f = family(row["Kernel_Name"])
        if f is None:
            continue
        a = acc[f]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
        a[2] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for f in sorted(set(fetch) | set(write)):
    n = fetch[f][0] or write[f][0]
    rd = 2.0 * fetch[f][1] * 1024 / max(1, fetch[f][0])      # bytes / launch, gfx950 x2 correction
    wr = write[f][1] * 1024 / max(1, write[f][0])
    out[f] = {"launches": n, "hbm_read_bytes_per_launch": round(rd), "hbm_write_bytes_per_launch": round(wr),
              "hbm_bytes_per_launch": round(rd + wr), "avg_ns_profiled": round(fetch[f][2] / max(1, fetch[f][0]))}
gem = [v for k, v in out.items() if k.startswith("gemm")]
tot_l = sum(v["launches"] for v in gem)
out["gemm_f16_kernel(all)"] = {
    "launches": tot_l,
    "hbm_bytes_per_launch": round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in gem) / tot_l),
    "hbm_read_bytes_per_launch": round(sum(v["hbm_read_bytes_per_launch"] * v["launches"] for v in gem) / tot_l),
    "hbm_write_bytes_per_launch": round(sum(v["hbm_write_bytes_per_launch"] * v["launches"] for v in gem) / tot_l),
}
# the kernel source this summary was captured on: bench.py quotes the summary only while csrc/gemm.hip is unchanged
import hashlib, os
_src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynamicscaler_amd", "csrc", "gemm.hip")
out["gemm_hip_sha256"] = hashlib.sha256(open(_src, "rb").read()).hexdigest()
# the residual mode of the PROFILED run: read from its bench.py JSON line (optional 4th argument: the run's stdout log, where
# config.residual_mode says what ran -- `--residual` on the command line or the library default); never from this process's environment
def _mode_of(log):
    try:
        for ln in open(log, errors="replace"):
            if ln.lstrip().startswith("{") and '"residual_mode"' in ln:
                return json.loads(ln)["config"]["residual_mode"]
    except Exception:
        pass
    return None


out["residual_mode"] = (_mode_of(sys.argv[4]) if len(sys.argv) > 4 else None) or "f32outer"
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
