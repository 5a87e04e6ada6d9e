#!/bin/bash
# round 6, call B: resident GroupNorm -- kernel tests, A/B per shape, step A/B (tune build), forced 256x320 for N = 1280
O=gpurun_out/r6_b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "groupnorm or layernorm" 2>&1 | tail -5 | tee $O/gn_tests.txt
timeout 900 python tools/bench_gn_resident.py 16 2 > $O/gn_ab.txt 2>&1
export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_tune.so
for rep in 1 2; do
  for m in 1 0; do
    DS_GN_RESIDENT=$m timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-configs 0 --wide-step 0 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('cfg3 DS_GN_RESIDENT=$m', d['ms_per_step'], d['result_sha256'])" | tee -a $O/step_ab.txt
  done
done
for m in 1 0; do
  DS_GN_RESIDENT=$m timeout 600 python bench.py --config col2 --tile-batch 1 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-configs 0 --wide-step 0 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('col2 tb1 DS_GN_RESIDENT=$m', d['ms_per_step'])" | tee -a $O/step_ab.txt
done
unset DS_HIP_LIBRARY
timeout 600 python tools/bench_tile_choice.py 16 > $O/tiles_e16.txt 2>&1
timeout 600 python tools/bench_tile_choice.py 8 > $O/tiles_e8.txt 2>&1
cat $O/gn_ab.txt $O/step_ab.txt; grep "L1280" $O/tiles_e16.txt $O/tiles_e8.txt | cut -c1-230
