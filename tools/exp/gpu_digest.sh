#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
O=gpurun_out/digest; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_multirank.py -q -x -k "bench_self_launch" 2>&1 | tail -5 | tee $O/tests.txt
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/a.json 2> $O/a.err
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --streams 1 --tile-batch 2 --graph 0 > $O/b.json 2> $O/b.err
grep -h -o '"result_sha256": {[^}]*}' $O/a.json $O/b.json | tee $O/summary.txt
