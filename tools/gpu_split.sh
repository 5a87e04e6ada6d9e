#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# a rank's share of an 8-GPU cfg3 step (one tile per level) emulated on one GPU with two columns (twice the work):
# one [cond | uncond] batch on one stream against cond / uncond on two streams
O=gpurun_out/split; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 env DS_SPLIT_CFG=2 python -m pytest tests/test_gpu_unet.py -q -x -k "pipelines_small or ring_pipeline_cfg_prefix" 2>&1 | tail -2 | tee $O/tests.txt
for rep in 1 2; do
  timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --streams 1 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/a_$rep.json 2> $O/a_$rep.err
  DS_SPLIT_CFG=2 timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --streams 2 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/b_$rep.json 2> $O/b_$rep.err
  timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --streams 1 --tile-batch 1 --share-cfg-prefix 0 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/c_$rep.json 2> $O/c_$rep.err
done
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob
O=sys.argv[1]
for k,name in (("a","one stream, [cond|uncond] batch per tile"),("b","cond / uncond on two streams"),("c","one stream, batch, no shared prefix")):
    v=[]
    for f in sorted(glob.glob(f"{O}/{k}_[0-9].json")):
        try: v.append(round(json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"],1))
        except Exception as e: v.append(open(f.replace(".json",".err")).read()[-200:])
    print(name, v, "-> per rank of 8:", [round(x/2,1) if isinstance(x,float) else x for x in v])
PY
