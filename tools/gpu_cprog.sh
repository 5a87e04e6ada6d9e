#!/bin/bash
# the C UNet program + everything that depends on the C packer: the whole GPU suite, then the default bench
O=gpurun_out/${1:-cprog}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_unet_c.py tests/test_gpu_handlers.py -m gpu -q -x -s > $O/gputest_c.log 2>&1; echo "c-program tests rc=$?" | tee -a $O/summary.txt
grep -h "rel err\|passed\|failed\|Error\|error" $O/gputest_c.log | tail -15 | tee -a $O/summary.txt
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -5 $O/gputest.log | tee -a $O/summary.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/summary.txt
tail -1 $O/bench.log | cut -c1-400 | tee -a $O/summary.txt
