#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# strict-precision mode: the new kernel forms + the strict UNet tests, then cfg3 bench in both residual-stream modes
O=gpurun_out/${1:-strict}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_strict.py -m gpu -q -x -s -k "fp32 or strict" > $O/gputest.log 2>&1; echo "gputest rc=$?" | tee -a $O/summary.txt
grep -h "rel err\|'test'\|passed\|failed\|Error" $O/gputest.log | tail -30 | tee -a $O/summary.txt
for rd in f16 f32; do
  DS_RESIDUAL_DTYPE=$rd timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_$rd.log 2>&1; echo "bench $rd rc=$?" | tee -a $O/summary.txt
  tail -1 $O/bench_$rd.log | cut -c1-600 | tee -a $O/summary.txt
done
