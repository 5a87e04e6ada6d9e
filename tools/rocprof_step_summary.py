#!/usr/bin/env python3
"""Summarise a `rocprofv3 --kernel-trace --stats` kernel_stats CSV of a bench.py run per kernel family.
usage: python tools/rocprof_step_summary.py kernel_stats.csv STEPS_IN_TRACE out.json
(STEPS_IN_TRACE = warmup + timed steps of the profiled bench.py command; run it with --no-roofline --no-cpu-baseline)."""
import csv, json, re, sys

FAMILIES = [("gemm", r"gemm_f16_kernel"), ("attention", r"attention_kernel"), ("temporal_attention", r"temporal_attention"),
            ("groupnorm", r"gn_"), ("layernorm", r"layernorm"), ("concat", r"concat_kernel|CatArray"),
            ("tile_ops", r"ring_gather|ring_scatter|renoise_mix|cfg_ddim"), ("misc", r"im2col|rows_to_ncthw|timestep_embedding|silu_kernel")]


def fam(name):
    if "temporal_attention" in name:
        return "temporal_attention"
    for f, pat in FAMILIES:
        if re.search(pat, name):
            return f
    return "other"


rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
agg = {}
tot = 0.0
for r in rows:
    f = fam(r["Name"])
    a = agg.setdefault(f, {"calls": 0, "ns": 0.0})
    a["calls"] += int(r["Calls"])
    a["ns"] += float(r["TotalDurationNs"])
    tot += float(r["TotalDurationNs"])
out = {"steps_in_trace": steps, "kernel_ms_per_step": tot / steps / 1e6, "families": {}}
for f, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    out["families"][f] = {"launches_per_step": a["calls"] / steps, "ms_per_step": a["ns"] / steps / 1e6,
                          "share": a["ns"] / tot, "avg_launch_us": a["ns"] / a["calls"] / 1e3}
fm = out["families"]
out["norms_and_concat_share"] = sum(fm.get(k, {}).get("share", 0.0) for k in ("groupnorm", "layernorm", "concat"))
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
