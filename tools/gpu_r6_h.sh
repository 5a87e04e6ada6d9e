#!/bin/bash
O=gpurun_out/r6_h; mkdir -p $O
export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_tune.so
for t in 2 3 1; do DS_GEMM_TILE=$t timeout 300 python tools/exp/coresidency.py 2>&1 | grep "^copy" | sed "s/^/tile $t: /" | tee -a $O/coresidency.txt; done
