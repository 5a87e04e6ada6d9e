#!/bin/bash
# the environment switches swept here exist only in the "tune" build variant (csrc/common.h DS_TUNING_ENV): build it on the box, load it
python -m dynamicscaler_amd.build --variant tune > /dev/null && export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_tune.so
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# GroupNorm at an 8-GPU rank's batch sizes: rows in flight per thread chosen by grid size (bit-neutral).  Micro-bench A/B
# (DS_GN_SPARSE_WGS=0 = the dense-grid variant everywhere), the invariance tests, the rank-share step, the default step.
O=gpurun_out/${1:-gnsparse}; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "groupnorm or gn" 2>&1 | tail -2 | tee $O/tests.txt
DS_GN_SPARSE_WGS=0 timeout 600 python tools/bench_norms.py 2 1 16 > $O/norms_dense.txt 2>&1
timeout 600 python tools/bench_norms.py 2 1 16 > $O/norms_sparse.txt 2>&1
for rep in 1 2; do
  DS_GN_SPARSE_WGS=0 timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --streams 1 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/share_dense_$rep.json 2> $O/share_dense_$rep.err
  timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --streams 1 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/share_sparse_$rep.json 2> $O/share_sparse_$rep.err
  DS_SPLIT_CFG=2 timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --streams 2 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/share2s_sparse_$rep.json 2> $O/share2s_sparse_$rep.err
done
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/default.json 2> $O/default.err
for f in $O/share_dense_*.json $O/share_sparse_*.json $O/share2s_sparse_*.json $O/default.json; do echo "$(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"latent_after_timed_steps": "[0-9a-f]*"' $f)"; done | tee $O/summary.txt
