#!/bin/bash
# the environment switches swept here exist only in the "tune" build variant (csrc/common.h DS_TUNING_ENV): build it on the box, load it
python -m dynamicscaler_amd.build --variant tune > /dev/null && export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_tune.so
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# A/B of an environment switch on one box: alternating short cfg3 bench runs.   usage: tools/gpu_env_ab.sh <tag> <VAR> <value> [<value> ...]
S=$1; VAR=$2; shift 2; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2 3; do
  for V in "$@"; do
    env $VAR=$V timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 ${BENCH_ARGS} > $O/${V}_$rep.json 2> $O/${V}_$rep.err
  done
done
python - $O "$@" <<'PY' | tee $O/summary.txt
import json,sys,glob
O=sys.argv[1]
for k in sys.argv[2:]:
    v=[json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"{O}/{k}_[0-9].json"))]
    print(f"{k:10s}", [round(x["ms_per_step"],1) for x in v], v[0]["result_sha256"] if v else None)
PY
