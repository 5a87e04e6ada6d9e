#!/bin/bash
# round 6, call L: final checks of bench.py as the driver runs it (--steps 20 --warmup 3), a run longer than the schedule, cfg5 eager on one
# stream (the command whose PMC pass crashed inside rocprofv3), the two-rank rehearsal
O=gpurun_out/r6_l; mkdir -p $O
pick() { python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('$1', d['ms_per_step'], d['config'].get('timed_region_starts_at_step'), d['result_sha256'], (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('traffic'))"; }
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --full-panorama 0 --streams 1 --graph 0 2> $O/cfg5_eager.err | pick "cfg5 eager 1 stream" | tee -a $O/summary.txt; echo "rc=${PIPESTATUS[0]}" | tee -a $O/summary.txt
( time timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_driver_like.json 2> $O/bench_driver_like.err ) 2> $O/time1.txt; echo "driver-like rc=$? lines=$(wc -l < $O/bench_driver_like.json) $(grep real $O/time1.txt)" | tee -a $O/summary.txt; cat $O/bench_driver_like.json | pick "driver-like" | tee -a $O/summary.txt
timeout 900 python bench.py --steps 60 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-configs 0 --wide-step 0 2> $O/long.err | pick "60 steps" | tee -a $O/summary.txt
timeout 2400 python -m pytest tests/test_gpu_multirank.py -q -x 2>&1 | tail -3 | tee -a $O/summary.txt
