#!/bin/bash
O=gpurun_out/norms; mkdir -p $O
for v in "" gnu8 gnu2; do
  if [ -n "$v" ]; then export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_$v.so; else unset DS_HIP_LIBRARY; fi
  python tools/bench_norms.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
done
