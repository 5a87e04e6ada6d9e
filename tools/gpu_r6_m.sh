#!/bin/bash
O=gpurun_out/r6_m; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_multirank.py -q 2>&1 | tail -4 | tee $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200 | tee -a $O/summary.txt
