#!/bin/bash
# round 3: GroupNorm kernel form chosen from the instance shape only (batch invariance at full size) -- tests, then the
# small-batch timing (8-GPU rank share) and the default bench on the new build
O=gpurun_out/gninv; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_unet_c.py -q -x 2>&1 | tail -3 | tee $O/tests_a.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "batch_equals or ring_pipeline or t24" 2>&1 | tail -3 | tee $O/tests_b.txt
for rep in 1 2; do
  timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/tb1_$rep.json 2> $O/tb1_$rep.err
  timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $O/def_$rep.json 2> $O/def_$rep.err
done
grep -h -o '"ms_per_step": [0-9.]*' $O/tb1_*.json $O/def_*.json | tee $O/summary.txt
