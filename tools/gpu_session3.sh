#!/bin/bash
# hazard: round 1's exact withdrawn GEMM (commit 67b5158, built in the build container from git history) under the guard
S=${1:-s3}; O=gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
run_guard() {  # $1 label
  for rep in 1 2 3; do
    hs=""
    for k in 1 2 3 4 5; do
      d=$(mktemp -d)
      timeout 300 python tests/multirank_worker.py $d grid4x2 reference > /dev/null 2>> $O/guard.err || echo "worker failed" >> $O/guard.err
      h=$(python -c "import numpy as np,hashlib,sys; o=np.load('$d/rank0.npz'); print(hashlib.sha256(o['den'].tobytes()+o['final'].tobytes()).hexdigest()[:10])")
      hs="$hs $h"
    done
    echo "$1 rep$rep:$hs" | tee -a $O/summary.txt
  done
}
export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_r1accinit.so
export DS_EXP_SHARED_BIAS_ROWS_MAX=1
# (the lazy-prepare switch of that session was removed afterwards) run_guard "r1accinit lazy-prepare(r1)"
unset DS_EXP_ROUND1_LAZY_PREPARE; run_guard "r1accinit fixed-prepare"
# the pytest guard itself (parent holds a GPU context), both modes
export DS_EXP_ROUND1_LAZY_PREPARE=1
for k in 1 2; do timeout 600 python -m pytest tests/test_gpu_multirank.py -q -k repeatable 2>&1 | tail -2 | tee -a $O/summary.txt; done
unset DS_EXP_ROUND1_LAZY_PREPARE
for k in 1 2; do timeout 600 python -m pytest tests/test_gpu_multirank.py -q -k repeatable 2>&1 | tail -2 | tee -a $O/summary.txt; done
timeout 600 python tests/hazard_probe.py unet 60 > $O/unet_probe.log 2>&1; grep '^{' $O/unet_probe.log | head -8 | tee -a $O/summary.txt
timeout 300 python tests/hazard_probe.py poison 2>&1 | grep '^{' | tee -a $O/summary.txt
