#!/bin/bash
O=gpurun_out/r6_r; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_handlers.py -q -x -k "sphere or stage_chain or handler" 2>&1 | tail -3 | tee $O/summary.txt
timeout 900 python tools/bench_sphere.py --model i2v --steps 4 2>&1 | tail -1 | cut -c1-400 | tee -a $O/summary.txt
