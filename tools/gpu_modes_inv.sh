#!/bin/bash
O=gpurun_out/modesinv; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -k "batch_equals or independent_of" 2>&1 | tail -15 | tee $O/tests.txt
