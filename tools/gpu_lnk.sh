#!/bin/bash
O=gpurun_out/lnk; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -q -x -s -k "layernorm" 2>&1 | grep -v Warning | tail -25 | tee $O/tests.txt
for rep in 1 2 3; do
  for V in 0 1 2; do
    DS_FOLD_LN=$V timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/f${V}_$rep.json 2> $O/f${V}_$rep.err
  done
done
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob
O=sys.argv[1]
for k in ("f0","f1","f2"):
    v=[]
    for f in sorted(glob.glob(f"{O}/{k}_[0-9].json")):
        try: v.append(round(json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"],1))
        except Exception as e: v.append(str(e)[:40])
    print("DS_FOLD_LN="+k[1], v)
PY
