#!/bin/bash
O=gpurun_out/group; mkdir -p $O
for g in 0 4 6 8 0 6; do
  DS_GEMM_GROUP_M=$g python tools/bench_wide_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
done
