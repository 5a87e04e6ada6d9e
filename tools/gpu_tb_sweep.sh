#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# what a rank of an N-GPU run would execute per level, measured on one GPU: cfg3 step time as a function of the tile batch
# (tiles per evaluation batch) and the number of streams -- the basis of the predicted strong-scaling curve (notes section 5)
O=gpurun_out/tbsweep; mkdir -p $O
for cfg in "1 8" "2 8" "1 4" "2 4" "1 2" "2 2" "1 1" "2 1"; do
  set -- $cfg
  timeout 600 python bench.py --steps 6 --warmup 3 --streams $1 --tile-batch $2 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/b_s$1_tb$2.json 2> $O/b_s$1_tb$2.err
  python - $O/b_s$1_tb$2.json "$1" "$2" <<'PY' | tee -a $O/summary.txt
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("streams",sys.argv[2],"tile_batch",sys.argv[3],"ms/step",round(j["ms_per_step"],1))
PY
done
