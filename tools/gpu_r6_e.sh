#!/bin/bash
# round 6, call E: loop tests with per-window stats, extended access-pattern microbenchmark, default bench line (stdout = one line), col2 rank share
O=gpurun_out/r6_e; mkdir -p $O
tools/exp/_build/access_patterns > $O/patterns.txt 2>&1
rm -f gpurun_out/measured_parity.jsonl
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -x -k "real_unet or real_i2v" 2>&1 | tail -3 | tee $O/loop_tests.txt
cp gpurun_out/measured_parity.jsonl $O/ 2>/dev/null
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$? lines=$(wc -l < $O/bench_default.json) $(grep real $O/bench_time.txt)" | tee $O/summary.txt
B="--steps 4 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-configs 0 --wide-step 0"
timeout 600 python bench.py --config col2 --tile-batch 1 $B > $O/col2_tb1_2streams.json 2>/dev/null
timeout 600 python bench.py --config col2 --tile-batch 1 --streams 1 $B > $O/col2_tb1_1stream.json 2>/dev/null
python -c "
import json
for f in ('col2_tb1_2streams','col2_tb1_1stream'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], 'per rank-step', d['ms_per_step']/2)
" | tee -a $O/summary.txt
cat $O/patterns.txt | head -8
