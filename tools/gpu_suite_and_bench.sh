#!/bin/bash
# quick check of a kernel change: GPU suite + cfg3 bench with the per-shape breakdown
S=${1:-s6}; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "gputest rc=$?" | tee -a $O/summary.txt
tail -5 $O/gputest.log | tee -a $O/summary.txt
DS_BENCH_BREAKDOWN=$O/shape_breakdown.csv timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - $O/bench_cfg3.json <<'PY' | tee -a $O/summary.txt
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j["roofline"]
print("ms/step", round(j["ms_per_step"],1), "gemm TF", r["achieved"], "frac", r["frac"], "avg_us", r["avg_launch_us"], "share", r["gemm_time_share_of_step"])
PY
head -25 $O/shape_breakdown.csv | tee -a $O/summary.txt
