#!/bin/bash
# A/B of an environment switch on one box: alternating short cfg3 bench runs.   usage: tools/gpu_abenv.sh <tag> VAR=VALUE
S=$1; KV=$2; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 env $KV python -m pytest tests/test_gpu_unet.py -q -x -k "unet_full or batch_equals or cfg_pair or tiny" 2>&1 | tail -2 | tee $O/tests.txt
for rep in 1 2 3; do
  timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/base_$rep.json 2> $O/base_$rep.err
  timeout 600 env $KV python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/var_$rep.json 2> $O/var_$rep.err
done
python - $O "$KV" <<'PY' | tee $O/summary.txt
import json,sys,glob
O=sys.argv[1]
for k in ("base","var"):
    v=[json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob(f"{O}/{k}_[0-9].json"))]
    print(k if k=="base" else sys.argv[2], [round(x,1) for x in v])
PY
