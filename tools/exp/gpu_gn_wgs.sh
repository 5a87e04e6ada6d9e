#!/bin/bash
# the environment switches swept here exist only in the "tune" build variant (csrc/common.h DS_TUNING_ENV): build it on the box, load it
python -m dynamicscaler_amd.build --variant tune > /dev/null && export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_tune.so
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# default step against the launch size up to which GroupNorm keeps 16 rows in flight per thread (bit-neutral)
O=gpurun_out/${1:-gnwgs}; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2 3; do
  for w in 1024 0 4096 16384 1000000; do
    DS_GN_SPARSE_WGS=$w timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/d_w${w}_$rep.json 2> $O/d_w${w}_$rep.err
  done
done
for w in 1024 4096 16384; do
  DS_GN_SPARSE_WGS=$w timeout 600 python bench.py --config cfg2 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/cfg2_w${w}.json 2> $O/cfg2_w${w}.err
done
for f in $O/d_*.json $O/cfg2_*.json; do echo "$(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f)"; done | tee $O/summary.txt
