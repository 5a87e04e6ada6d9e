#!/bin/bash
# wide-mode check: kernel + UNet tests of the mode, ms per evaluation pair, rocprofv3 kernel stats of the evaluation
O=gpurun_out/${1:-widecheck}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_wide.py -q -x 2>&1 | tail -3 | tee $O/tests.txt
python tools/bench_wide.py 3 2>&1 | tail -2 | tee $O/wide_ms.txt
bash tools/gpu_wide_profile.sh ${1:-widecheck}/prof
