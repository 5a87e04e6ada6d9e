#!/bin/bash
O=gpurun_out/attn; mkdir -p $O
for v in "" attnplain "" attnplain; do
  if [ -n "$v" ]; then export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_$v.so; else unset DS_HIP_LIBRARY; fi
  python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
done
unset DS_HIP_LIBRARY
timeout 900 python -m pytest tests/test_gpu_multirank.py -q 2>&1 | tail -3 | tee -a $O/summary.txt
