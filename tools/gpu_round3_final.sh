#!/bin/bash
# end-of-round evidence: whole GPU suite, default bench (with roofline + cpu_baseline), the other BASELINE configs, the tile-batch sweep
O=gpurun_out/${1:-r3final}; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/gputest.log | tail -10 | tee -a $O/summary.txt
timeout 1200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log)" | tee -a $O/summary.txt
timeout 1500 python bench.py > $O/bench_default.log 2>&1; echo "bench default rc=$?" | tee -a $O/summary.txt
tail -1 $O/bench_default.log > $O/bench_default.json; cut -c1-300 $O/bench_default.json | tee -a $O/summary.txt
for cfg in cfg2 cfg4 cfg5; do
  timeout 1500 python bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --full-panorama 0 > $O/bench_$cfg.log 2>&1
  echo "bench $cfg rc=$?: $(tail -1 $O/bench_$cfg.log | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["roofline"]["achieved"])' 2>&1 | tail -1)" | tee -a $O/summary.txt
done
DS_RESIDUAL_DTYPE=f32 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --full-panorama 0 > $O/bench_strict.log 2>&1
echo "bench strict: $(tail -1 $O/bench_strict.log | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["roofline"]["achieved"])' 2>&1 | tail -1)" | tee -a $O/summary.txt
bash tools/gpu_tb_sweep.sh > /dev/null 2>&1; cat gpurun_out/tbsweep/summary.txt | tee -a $O/summary.txt
cp gpurun_out/measured_parity.jsonl $O/ 2>/dev/null
