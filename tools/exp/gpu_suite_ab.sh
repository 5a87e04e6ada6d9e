#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# the whole GPU suite (all failures listed), then bench alternating the C and the Python launch program
O=gpurun_out/${1:-suite}; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/gputest.log | tail -15 | tee -a $O/summary.txt
for rep in 1 2; do for prog in c python; do
  # (DS_UNET_PROGRAM is gone since round 5: one launch program, the C one)
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/bench_${prog}_$rep.log 2>&1
  echo "bench $prog $rep: $(tail -1 $O/bench_${prog}_$rep.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])' 2>&1 | tail -1)" | tee -a $O/summary.txt
done; done
