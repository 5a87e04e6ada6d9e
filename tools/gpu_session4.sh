#!/bin/bash
# round 2, session 4: full GPU suite (verbose), benches (cfg3 + tile-batch sweep + cfg2/4/5), rocprofv3 kernel stats
S=${1:-s4}; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -s -rA > $O/gputest_verbose.log 2>&1; echo "gputest rc=$?" | tee -a $O/summary.txt
tail -4 $O/gputest_verbose.log | tee -a $O/summary.txt
grep -h "FAILED\|Error" $O/gputest_verbose.log | head -20 | tee -a $O/summary.txt
export DS_HIP_LIBRARY=$R/dynamicscaler_amd/libdynscaler_hip_barebarrier.so
timeout 600 python tests/hazard_probe.py unet 100 2>&1 | grep '^{' | head -6 | tee -a $O/summary.txt
unset DS_HIP_LIBRARY
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench cfg3 rc=$?" | tee -a $O/summary.txt
for tb in 1 2 4; do
  timeout 600 python bench.py --steps 4 --warmup 2 --tile-batch $tb --no-cpu-baseline --no-roofline > $O/bench_tb$tb.json 2> $O/bench_tb$tb.err
  timeout 600 python bench.py --steps 4 --warmup 2 --tile-batch $tb --streams 1 --no-cpu-baseline --no-roofline > $O/bench_tb${tb}_s1.json 2> $O/bench_tb${tb}_s1.err
done
timeout 600 python bench.py --steps 4 --warmup 2 --streams 1 --graph 0 --no-cpu-baseline --no-roofline > $O/bench_serial_eager.json 2> $O/bench_serial_eager.err
timeout 900 python bench.py --config cfg2 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?" | tee -a $O/summary.txt
timeout 900 python bench.py --config cfg4 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?" | tee -a $O/summary.txt
timeout 1500 python bench.py --config cfg5 --steps 3 --warmup 1 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?" | tee -a $O/summary.txt
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
  head -12 $O/kernel_stats_$c.csv | cut -c1-200 | tee -a $O/summary.txt
done
du -sh $O | tee -a $O/summary.txt
