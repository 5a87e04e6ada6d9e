#!/bin/bash
# hazard: the round-1 guard workload (5 fresh processes, toy ring, 2 streams x graph replays) under the accinit diagnostic
# build, with this round's prepare()/sync fix and with round 1's lazy prepare restored.
S=${1:-s2}; O=gpurun_out/$S; mkdir -p $O
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
export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_accinit.so
unset DS_EXP_ROUND1_LAZY_PREPARE; run_guard "accinit fixed-prepare"
# (the lazy-prepare switch of that session was removed afterwards) run_guard "accinit lazy-prepare(r1)"
unset DS_HIP_LIBRARY
run_guard "product lazy-prepare(r1)"
unset DS_EXP_ROUND1_LAZY_PREPARE; run_guard "product fixed-prepare"
export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_accinit.so
timeout 600 python tests/hazard_probe.py unet 60 > $O/unet_probe.log 2>&1; grep '^{' $O/unet_probe.log | head -8 | tee -a $O/summary.txt
unset DS_HIP_LIBRARY
timeout 600 python tests/hazard_probe.py unet 60 > $O/unet_probe_p.log 2>&1; grep '^{' $O/unet_probe_p.log | head -8 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "temporal_attention" 2>&1 | tail -3 | tee -a $O/summary.txt
timeout 1200 python -m pytest tests -m gpu -q -s -rA > $O/gputest_verbose.log 2>&1; echo "gputest rc=$?" | tee -a $O/summary.txt
tail -3 $O/gputest_verbose.log | tee -a $O/summary.txt
