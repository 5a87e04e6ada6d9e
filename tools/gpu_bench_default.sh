#!/bin/bash
O=gpurun_out/${1:-bdef}; mkdir -p $O
T0=$(date +%s); python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$? wall $(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
python - $O/bench.json <<'PY' | tee -a $O/summary.txt
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print({k: j[k] for k in ("value","ms_per_step","steps","warmup","sec_per_50_step_panorama","sec_per_50_step_panorama_is","speedup_vs_cpu_baseline")}); print(j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["rocprof"], j["cpu_baseline"])
PY
