#!/bin/bash
# round 6, call C: tile rule + resident GroupNorm A/B in the step and in the sphere stage; g41 test; default bench line with gen_pano_360_default
O=gpurun_out/r6_c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg5_dependency_chain or cfg3_headline" 2>&1 | tail -4 | tee $O/g41_test.txt
export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_tune.so
B="--steps 4 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-configs 0 --wide-step 0"
pick() { python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('$1', d['ms_per_step'], d['result_sha256'])"; }
for rep in 1 2; do
  DS_GEMM_PREF320=1 DS_GN_RESIDENT=1 timeout 600 python bench.py $B 2>/dev/null | pick "cfg3 pref320=1 gnres=1" | tee -a $O/step_ab.txt
  DS_GEMM_PREF320=0 DS_GN_RESIDENT=1 timeout 600 python bench.py $B 2>/dev/null | pick "cfg3 pref320=0 gnres=1" | tee -a $O/step_ab.txt
  DS_GEMM_PREF320=1 DS_GN_RESIDENT=0 timeout 600 python bench.py $B 2>/dev/null | pick "cfg3 pref320=1 gnres=0" | tee -a $O/step_ab.txt
done
for m in 1 0; do
  DS_GN_RESIDENT=$m timeout 600 python bench.py --config col2 --tile-batch 1 --streams 1 $B 2>/dev/null | pick "col2 tb1 1stream gnres=$m" | tee -a $O/step_ab.txt
  DS_GEMM_PREF320=$m timeout 600 python bench.py --config col2 --tile-batch 1 --streams 1 $B 2>/dev/null | pick "col2 tb1 1stream pref320=$m" | tee -a $O/step_ab.txt
  DS_GN_RESIDENT=$m timeout 900 python tools/bench_sphere.py --model t2v --steps 3 2>&1 | tail -2 | sed "s/^/sphere t2v gnres=$m: /" | tee -a $O/step_ab.txt
done
unset DS_HIP_LIBRARY
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 6000 $O/bench_default.json
