#!/bin/bash
# the whole GPU suite with the measured numbers kept (-s output), smoke(), the default bench line
O=gpurun_out/${1:-suite}; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 4000 python -m pytest tests -m gpu -q -s > $O/gputest.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
grep -h "^FAILED\|^ERROR\| passed\| failed" $O/gputest.log | tail -10 | tee -a $O/summary.txt
grep -h "rel err\|rel_err" $O/gputest.log | cut -c1-200 > $O/measured_lines.txt
timeout 1200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log)" | tee -a $O/summary.txt
python tools/bench_wide.py 3 2>&1 | tail -2 | tee -a $O/summary.txt
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$? $(grep real $O/bench_time.txt)" | tee -a $O/summary.txt
cut -c1-250 $O/bench_default.json | tee -a $O/summary.txt
cp gpurun_out/measured_parity.jsonl $O/ 2>/dev/null
