#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# A/B of build variants against the product library on one box: alternating short cfg3 bench runs, then one run each with
# the per-shape breakdown.   usage: tools/gpu_ab.sh <tag> <variant> [<variant> ...]
S=$1; shift; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2 3; do
  for V in base "$@"; do
    if [ $V = base ]; then unset DS_HIP_LIBRARY; else export DS_HIP_LIBRARY=$R/dynamicscaler_amd/libdynscaler_hip_$V.so; fi
    timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/${V}_$rep.json 2> $O/${V}_$rep.err
  done
done
for V in base "$@"; do
  if [ $V = base ]; then unset DS_HIP_LIBRARY; else export DS_HIP_LIBRARY=$R/dynamicscaler_amd/libdynscaler_hip_$V.so; fi
  DS_BENCH_BREAKDOWN=$O/shape_$V.csv timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --full-panorama 0 > $O/r_$V.json 2> $O/r_$V.err
done
python - $O base "$@" <<'PY' | tee $O/summary.txt
import json,sys,glob,csv
O=sys.argv[1]; V=sys.argv[2:]
for k in V:
    v=[json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob(f"{O}/{k}_[0-9].json"))]
    r=json.loads(open(f"{O}/r_{k}.json").read().strip().splitlines()[-1])["roofline"]
    print(f"{k:8s}", [round(x,1) for x in v], "gemm TF", r["achieved"], "avg_us", r["avg_launch_us"])
t={k:{(r['kernel'],r['shape']):r for r in csv.DictReader(open(f"{O}/shape_{k}.csv"))} for k in V}
b=t["base"]
for key in sorted(b, key=lambda k:-float(b[k]['ms_per_step']))[:24]:
    print(f"{key[0]:10s} {key[1]:28s}", "  ".join(f"{k} {float(t[k][key]['ms_per_step']):6.2f}" for k in V if key in t[k]))
PY
