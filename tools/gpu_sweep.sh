#!/bin/bash
O=gpurun_out/sweep; mkdir -p $O
for cfg in "2 8" "3 8" "4 8" "2 4" "4 4" "1 8"; do
  set -- $cfg
  timeout 600 python bench.py --steps 6 --warmup 3 --streams $1 --tile-batch $2 --no-cpu-baseline --no-roofline > $O/b_s$1_tb$2.json 2> $O/b_s$1_tb$2.err
  python - $O/b_s$1_tb$2.json "$1" "$2" <<'PY' | tee -a $O/summary.txt
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("streams",sys.argv[2],"tile_batch",sys.argv[3],"ms/step",round(j["ms_per_step"],1))
PY
done
