#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# batched crop embedding of the i2v pipelines: encoder + i2v pipeline tests, then cfg4 with the default warmup and at steady state
O=gpurun_out/${1:-crops}; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_encoders.py -q -x 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $O/tests.txt
timeout 1500 python -m pytest tests -m gpu -q -x -k "i2v" 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee -a $O/tests.txt
timeout 1200 python bench.py --config cfg4 --no-cpu-baseline --no-roofline > $O/cfg4_default.json 2> $O/cfg4_default.err
timeout 1200 python bench.py --config cfg4 --steps 6 --warmup 10 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/cfg4_steady.json 2> $O/cfg4_steady.err
for f in $O/cfg4_*.json; do echo "$(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"sec_per_50_step_panorama": [0-9.]*' $f) $(grep -o '"result_sha256": {[^}]*}' $f)"; done | tee $O/summary.txt
