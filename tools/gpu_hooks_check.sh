#!/bin/bash
# the C launch program's instrumentation hooks: poison run, layer-wise error budget (block taps), bench roofline through the launch hook
O=gpurun_out/${1:-hooks}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -s -k "uninitialised_cu_state or error_budget" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 | tee $O/tests.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --full-panorama 0 --other-configs 0 --wide-step 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/tests.txt
python - <<'PY' | tee -a $O/tests.txt
import json
j=json.loads(open("gpurun_out/'"${1:-hooks}"'/bench.json").read().strip().splitlines()[-1])
r=j["roofline"]; print(j["ms_per_step"], {k:r[k] for k in ("achieved","frac","launches_per_step","avg_launch_us","algorithmic_tflop_per_step","attention_tflops","gemm_time_share_of_step")})
PY
