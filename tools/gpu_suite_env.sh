#!/bin/bash
# the whole GPU suite under an environment switch.   usage: tools/gpu_suite_env.sh <tag> VAR=VALUE
O=gpurun_out/$1; mkdir -p $O
timeout 1800 env $2 python -m pytest tests -m gpu -q -s > $O/gputest.log 2>&1; echo "gputest rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed" $O/gputest.log | tail -3 | tee -a $O/summary.txt
grep -E "^FAILED|Error|assert" $O/gputest.log | head -20 | tee -a $O/summary.txt
