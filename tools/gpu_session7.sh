#!/bin/bash
# after the register epilogue: suite, cfg3 bench + breakdowns (tile batch 8 and 1), cfg4, rocprof kernel stats cfg3 / cfg5
S=${1:-s7}; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "gputest rc=$?" | tee -a $O/summary.txt
tail -5 $O/gputest.log | tee -a $O/summary.txt
DS_BENCH_BREAKDOWN=$O/shape_breakdown_tb8.csv timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench rc=$?" | tee -a $O/summary.txt
DS_BENCH_BREAKDOWN=$O/shape_breakdown_tb1.csv timeout 900 python bench.py --steps 4 --warmup 2 --tile-batch 1 --streams 1 --no-cpu-baseline > $O/bench_tb1_s1.json 2> $O/bench_tb1_s1.err
timeout 900 python bench.py --steps 4 --warmup 2 --tile-batch 1 --no-cpu-baseline --no-roofline > $O/bench_tb1.json 2> $O/bench_tb1.err
timeout 900 python bench.py --config cfg4 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?" | tee -a $O/summary.txt
timeout 1500 python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?" | tee -a $O/summary.txt
for f in $O/bench_*.json; do python - "$f" <<'PY' | tee -a $O/summary.txt
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "ms/step", round(j["ms_per_step"],1), "value", round(j["value"],4), "gemm TF", r.get("achieved"), "frac", r.get("frac"))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unparsed", e)
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof_cfg3.log 2>&1; echo "rocprof cfg3 rc=$?" | tee -a $O/summary.txt
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $R/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof_cfg5.log 2>&1; echo "rocprof cfg5 rc=$?" | tee -a $O/summary.txt
cd $R
for c in cfg3 cfg5; do
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$c.csv
  find $O/prof_$c -name "*kernel_trace.csv" -delete; find $O/prof_$c -name "*.db" -delete
  head -8 $O/kernel_stats_$c.csv | cut -c1-200 | tee -a $O/summary.txt
done
head -30 $O/shape_breakdown_tb8.csv | tee -a $O/summary.txt
du -sh $O | tee -a $O/summary.txt
