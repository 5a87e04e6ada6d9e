#!/bin/bash
O=gpurun_out/${1:-strictplus}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_strict.py tests/test_gpu_schedule50.py tests/test_gpu_unet_c.py tests/test_gpu_fullsize.py -m gpu -q -s -k "strict or float32 or ring_pipeline_with or c_program" > $O/t.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
grep -h "full_strict\|worst\|free_running\|ring_real\|tiny_strict\|toy \|passed\|failed" $O/t.log | cut -c1-420 | tee -a $O/summary.txt
for rep in 1 2; do for rd in f16 f32; do
  DS_RESIDUAL_DTYPE=$rd timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/bench_${rd}_$rep.log 2>&1
  echo "bench $rd $rep: $(tail -1 $O/bench_${rd}_$rep.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])' 2>&1 | tail -1)" | tee -a $O/summary.txt
done; done
cp gpurun_out/measured_parity.jsonl $O/ 2>/dev/null
