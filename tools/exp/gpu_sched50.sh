#!/bin/bash
# the 50-step schedule parity tests (teacher-forced per index + free-running drift), plus the tests fixed after the last suite run
O=gpurun_out/${1:-sched50}; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_schedule50.py -m gpu -q -s > $O/sched50.log 2>&1; echo "sched50 rc=$?" | tee -a $O/summary.txt
grep -h "'test'\|worst\|passed\|failed" $O/sched50.log | cut -c1-700 | tee -a $O/summary.txt
timeout 1200 python -m pytest tests/test_gpu_handlers.py tests/test_gpu_multirank.py "tests/test_gpu_unet.py::test_unet_cfg_pair_prefix_sharing_is_bit_identical" "tests/test_gpu_unet.py::test_ring_pipeline_baseline_geometries_fake_eps_bit_exact" tests/test_gpu_fullsize.py -m gpu -q > $O/fixed.log 2>&1; echo "fixed tests rc=$?" | tee -a $O/summary.txt
tail -4 $O/fixed.log | tee -a $O/summary.txt
cp gpurun_out/measured_parity.jsonl $O/ 2>/dev/null
