#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# last check of a round: the whole GPU suite, smoke(), the default bench line
O=gpurun_out/${1:-finalcheck}; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/gputest.log | tail -10 | tee -a $O/summary.txt
timeout 1200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log)" | tee -a $O/summary.txt
timeout 1500 python bench.py > $O/bench_default.log 2>&1; echo "bench default rc=$?" | tee -a $O/summary.txt
tail -1 $O/bench_default.log > $O/bench_default.json; cut -c1-250 $O/bench_default.json | tee -a $O/summary.txt
cp gpurun_out/measured_parity.jsonl $O/ 2>/dev/null
