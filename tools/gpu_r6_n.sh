#!/bin/bash
O=gpurun_out/r6_n; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_strict.py -q -x -k "strict_twin or tiny_strict" 2>&1 | tail -15 | tee $O/summary.txt
