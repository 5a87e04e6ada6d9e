#!/bin/bash
# the environment switches swept here exist only in the "tune" build variant (csrc/common.h DS_TUNING_ENV): build it on the box, load it
python -m dynamicscaler_amd.build --variant tune > /dev/null && export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_tune.so
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# A/B of the GroupNorm chunk rule on one box: default step (tile batch 8, two streams) and the 8-GPU rank share
O=gpurun_out/${1:-gnrule}; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2 3; do
  for rule in 0 1 2; do
    DS_GN_CHUNK_RULE=$rule timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/default_rule${rule}_$rep.json 2> $O/default_rule${rule}_$rep.err
  done
done
for rule in 0 1 2; do
  DS_GN_CHUNK_RULE=$rule DS_SPLIT_CFG=2 timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --streams 2 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/share_rule${rule}.json 2> $O/share_rule${rule}.err
done
for f in $O/default_rule*.json $O/share_rule*.json; do echo "$(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f)"; done | tee $O/summary.txt
