#!/bin/bash
# quick GPU check: the whole GPU suite, stop at the first failure
O=gpurun_out/${1:-quick}; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x -s > $O/gputest.log 2>&1; echo "gputest rc=$?" | tee -a $O/summary.txt; tail -4 $O/gputest.log | tee -a $O/summary.txt
grep -h "get scale factor\|clear clip\|multi-prompt\|folded" $O/gputest.log | tee -a $O/summary.txt
