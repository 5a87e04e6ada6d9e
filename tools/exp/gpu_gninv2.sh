#!/bin/bash
# the batch-invariance tests against round 2's GroupNorm kernel choice (variant "gncount": must FAIL), then the 8-GPU rank share
O=gpurun_out/gninv2; mkdir -p $O
export PYTHONUNBUFFERED=1
DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_gncount.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -k "batch_invariant or batch_equals" 2>&1 | grep -E "^FAILED|passed|failed|AssertionError" | tee $O/old_form.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -k "batch_invariant or batch_equals" 2>&1 | grep -E "^FAILED|passed|failed" | tee $O/new_form.txt
tools/gpu_split.sh
cp gpurun_out/split/summary.txt $O/split_summary.txt
