#!/bin/bash
# GroupNorm / LayerNorm streaming rates of the product library (tools/bench_norms.py)
O=gpurun_out/norms; mkdir -p $O
python tools/bench_norms.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
