#!/usr/bin/env python3
"""Curated per-shape table from tools/pmc_sq_summary.py's raw summary of the tools/pmc_shapes.py launches
(profiles/r2_pmc_mfma_util.json).   usage: python tools/pmc_shapes_table.py raw_summary.json out.json "source note" """
import json, sys
raw = json.load(open(sys.argv[1]))
# (kernel key in the raw summary = template + grid size of the launch) -> (shape label, algorithmic FLOPs)
SHAPES = [
    ("attention_kernel grid=6553600", "self-attention 256 x 5 heads x 2560 x 2560", 4.0 * 256 * 5 * 2560 * 2560 * 64),
    ("gemm<256x256,mode0,ns2> grid=13107200", "GEGLU linear 655360x2560x320 (level-1 FF1)", 2.0 * 655360 * 2560 * 320),
    ("gemm<256x256,mode0,ns2> grid=6553600", "GEGLU linear 163840x5120x640 (level-2 FF1)", 2.0 * 163840 * 5120 * 640),
    ("gemm<256x256,mode1,ns2> grid=409600", "conv3x3 40960x1280x11520 (level 3)", 2.0 * 40960 * 1280 * 11520),
    ("gemm<256x320,mode0,ns2> grid=1310720", "linear 655360x320x320 + bias + residual (level-1 to_out)", 2.0 * 655360 * 320 * 320),
    ("gemm<256x320,mode0,ns2> grid=3932160", "linear 655360x960x320 (level-1 QKV)", 2.0 * 655360 * 960 * 320),
    ("gemm<256x320,mode1,ns2> grid=1310720", "conv3x3 655360x320x2880 (level 1, taps outer)", 2.0 * 655360 * 320 * 2880),
    ("gemm<256x320,mode4,ns2> grid=1310720", "conv3x3 655360x320x2880 (level 1, taps innermost)", 2.0 * 655360 * 320 * 2880),
    ("gemm<256x320,mode2,ns2> grid=1310720", "temporal conv 655360x320x960 (level 1)", 2.0 * 655360 * 320 * 960),
]
out = {"_source": sys.argv[3],
       "_definitions": {"mfma_busy": "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): matrix-pipe busy cycles per CU-active cycle",
                        "valu_per_mfma": "(SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA", "clock_ghz": "GRBM_GUI_ACTIVE / 8 / duration",
                        "wait_any / wait_inst_any / active_inst_any / wait_inst_lds": "fraction of SQ_WAVE_CYCLES",
                        "lds_bank_conflict": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
                        "hbm_read_gb / hbm_write_gb": "FETCH_SIZE x 2 KiB-units / WRITE_SIZE KiB-units per launch (gfx950 correction of MI355X_MICROARCH.md)"},
       "shapes": {}}
for key, label, flops in SHAPES:
    r = raw.get(key)
    if not r:
        continue
    us = r["avg_us_profiled"]
    out["shapes"][label] = {
        "kernel": key, "us_per_launch_profiled": round(us, 1), "mfma_busy": round(r["mfma_busy_per_cu_cycle"], 3),
        "wait_any": round(r["SQ_WAIT_ANY_frac_of_wave_cycles"], 3), "wait_inst_any": round(r["SQ_WAIT_INST_ANY_frac_of_wave_cycles"], 3),
        "active_inst_any": round(r["SQ_ACTIVE_INST_ANY_frac_of_wave_cycles"], 3), "wait_inst_lds": round(r["SQ_WAIT_INST_LDS_frac_of_wave_cycles"], 3),
        "valu_per_mfma": round(r["valu_per_mfma"], 2), "lds_bank_conflict": round(r["lds_bank_conflict_frac"], 3),
        "clock_ghz": round(r["effective_clock_ghz"], 2), "hbm_read_gb": round(r["FETCH_SIZE"] * 2 * 1024 / 1e9, 3),
        "hbm_write_gb": round(r["WRITE_SIZE"] * 1024 / 1e9, 3), "tflops_profiled": round(flops / us / 1e6, 1)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out["shapes"].items():
    print(f"{k:58s} busy {v['mfma_busy']:.3f}  {v['tflops_profiled']:7.1f} TF  rd {v['hbm_read_gb']:.3f} wr {v['hbm_write_gb']:.3f} GB  clk {v['clock_ghz']}")
