#!/bin/bash
# the environment switches swept here exist only in the "tune" build variant (csrc/common.h DS_TUNING_ENV): build it on the box, load it
python -m dynamicscaler_amd.build --variant tune > /dev/null && export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_tune.so
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# rank-share step against the workgroup count from which choose_tile takes the one-workgroup-per-CU tiles (bit-neutral)
O=gpurun_out/${1:-bigmin}; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2; do
for bm in 160 257 330 520 100; do
  DS_GEMM_BIG_MIN=$bm timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --streams 1 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/s1_bm${bm}_$rep.json 2> $O/s1_bm${bm}_$rep.err
  DS_GEMM_BIG_MIN=$bm DS_SPLIT_CFG=2 timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --streams 2 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/s2_bm${bm}_$rep.json 2> $O/s2_bm${bm}_$rep.err
done
done
for f in $O/s1_*.json $O/s2_*.json; do echo "$(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"latent_after_timed_steps": "[0-9a-f]*"' $f)"; done | tee $O/summary.txt
