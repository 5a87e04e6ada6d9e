#!/bin/bash
# the "outer" precision mode (fp32 stream between the blocks only): strict + schedule tests, then bench in the three modes, alternating
O=gpurun_out/${1:-outer}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_strict.py tests/test_gpu_schedule50.py tests/test_gpu_unet_c.py -m gpu -q -s > $O/t.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
grep -h "full_strict\|worst\|free_running\|passed\|failed" $O/t.log | cut -c1-500 | tee -a $O/summary.txt
for rep in 1 2; do for rd in f16 f32outer f32; do
  DS_RESIDUAL_DTYPE=$rd timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/bench_${rd}_$rep.log 2>&1
  echo "bench $rd $rep: $(tail -1 $O/bench_${rd}_$rep.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])' 2>&1 | tail -1)" | tee -a $O/summary.txt
done; done
cp gpurun_out/measured_parity.jsonl $O/ 2>/dev/null
