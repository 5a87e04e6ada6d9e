#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# ms per cfg3 step over --streams x --tile-batch (default residual mode), two runs each.   usage: tools/gpu_sweep_streams.sh <tag>
S=$1; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2; do
  for cfg in "2 8" "1 8" "3 8" "2 4" "4 4"; do
    set -- $cfg
    timeout 600 python bench.py --steps 6 --warmup 2 --streams $1 --tile-batch $2 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 > $O/s$1_tb$2_$rep.json 2> $O/s$1_tb$2_$rep.err
  done
done
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
O=sys.argv[1]
for f in sorted(glob.glob(f"{O}/s*_tb*_1.json")):
    k=os.path.basename(f)[:-7]
    v=[json.loads(open(g).read().strip().splitlines()[-1])["ms_per_step"] for g in sorted(glob.glob(f"{O}/{k}_[0-9].json"))]
    print(k, [round(x,1) for x in v])
PY
