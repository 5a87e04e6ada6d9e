#!/bin/bash
# quick sanity of a rebuilt library: kernel + C-program tests, smoke()
O=gpurun_out/${1:-quick}; mkdir -p $O
timeout 1500 python -m pytest ${GPU_QUICK_TESTS:-tests/test_gpu_kernels.py tests/test_gpu_unet_c.py} -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/summary.txt
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
