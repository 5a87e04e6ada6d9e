#!/bin/bash
O=gpurun_out/${1:-r3a}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_schedule50.py -m gpu -q -s > $O/sched50.log 2>&1; echo "sched50 rc=$?" | tee -a $O/summary.txt
grep -h "worst\|free_running\|passed\|failed" $O/sched50.log | cut -c1-600 | tee -a $O/summary.txt
for lat in f32 f16 f32 f16; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 --latents $lat > $O/bench_$lat.log 2>&1
  echo "bench latents $lat: $(tail -1 $O/bench_$lat.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])' 2>&1 | tail -1)" | tee -a $O/summary.txt
done
cp gpurun_out/measured_parity.jsonl $O/ 2>/dev/null
