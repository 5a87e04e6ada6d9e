#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# the other BASELINE.json configurations on the final build
O=gpurun_out/${1:-configs}; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python bench.py --config cfg2 --steps 6 --warmup 2 --no-cpu-baseline > $O/cfg2.json 2> $O/cfg2.err
timeout 1500 python bench.py --config cfg4 --steps 6 --warmup 10 --no-cpu-baseline > $O/cfg4.json 2> $O/cfg4.err
timeout 1500 python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $O/cfg5.json 2> $O/cfg5.err
for m in f16 f32; do timeout 900 python bench.py --residual $m --other-mode 0 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/cfg3_$m.json 2> $O/cfg3_$m.err; done
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/cfg3.json 2> $O/cfg3.err
for f in $O/*.json; do echo "$(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"sec_per_50_step_panorama": [0-9.]*' $f) $(grep -o '"achieved": [0-9.]*' $f | head -1)"; done | tee $O/summary.txt
