#!/bin/bash
# final verification of a round: smoke, full GPU suite, default bench, kernel stats (1 stream and default), step HBM traffic
S=${1:-final}; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -2 $O/smoke.log | tee -a $O/summary.txt
timeout 1800 python -m pytest tests -m gpu -q -s -rA > $O/gputest_verbose.log 2>&1; echo "gputest rc=$?" | tee -a $O/summary.txt; tail -2 $O/gputest_verbose.log | tee -a $O/summary.txt
DS_BENCH_BREAKDOWN=$O/shape_breakdown_tb8.csv timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout 900 python bench.py --config cfg2 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 900 python bench.py --config cfg4 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 1500 python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
for f in $O/bench_*.json; do python - "$f" <<'PY' | tee -a $O/summary.txt
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "ms/step", round(j["ms_per_step"],1), "value", round(j["value"],4), "gemm TF", r.get("achieved"), "frac", r.get("frac"), "rocprof", (r.get("rocprof") or {}).get("frac"))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unparsed", e)
PY
done
bash tools/gpu_prof.sh $S
