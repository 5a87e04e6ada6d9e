#!/bin/bash
O=gpurun_out/conv; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or conv" 2>&1 | tail -3 | tee -a $O/summary.txt
for rep in 1 2; do
  echo "== taps inner" | tee -a $O/summary.txt
  python tools/bench_gemm.py 2>&1 | grep "conv3x3" | tee -a $O/summary.txt
  echo "== taps outer (round-1 order)" | tee -a $O/summary.txt
  DS_CONV_TAPS_OUTER=1 python tools/bench_gemm.py 2>&1 | grep "conv3x3" | tee -a $O/summary.txt
done
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "gputest rc=$?" | tee -a $O/summary.txt; tail -3 $O/gputest.log | tee -a $O/summary.txt
DS_BENCH_BREAKDOWN=$O/shape_breakdown.csv timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ti.json 2> $O/bench_ti.err
DS_CONV_TAPS_OUTER=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_to.json 2> $O/bench_to.err
for f in $O/bench_ti.json $O/bench_to.json; do python - $f <<'PY' | tee -a $O/summary.txt
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
print(sys.argv[1].split("/")[-1], "ms/step", round(j["ms_per_step"],1), "gemm TF", r.get("achieved"))
PY
done
grep "^gemm,1x" $O/shape_breakdown.csv | head -12 | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmcF -- python3 $R/tools/pmc_shapes.py > $R/$O/pmcF.log 2>&1
cd $R
ff=$(find $O/pmcF -name "*counter_collection.csv" | head -1)
python3 tools/pmc_sq_summary.py $O/pmc_fetch.json $ff > /dev/null 2>&1
python3 - $O/pmc_fetch.json <<'PY' | tee -a $O/summary.txt
import json,sys
j=json.load(open(sys.argv[1]))
for k,v in j.items():
    if 'mode' in k: print(k, "read GB/launch", round(2*v.get("FETCH_SIZE",0)*1024/1e9,3), "us", round(v["avg_us_profiled"],1))
PY
