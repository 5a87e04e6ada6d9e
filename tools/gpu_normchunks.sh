#!/bin/bash
O=gpurun_out/normchunks; mkdir -p $O
timeout 300 python tools/bench_norm_chunks.py 2>&1 | grep -v amdgpu.ids | tee $O/summary.txt
