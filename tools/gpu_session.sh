#!/bin/bash
# One gpurun call of round 2: parity at full size, hazard probes (product + accinit diagnostic build), GPU suite, bench.
# Usage (build container): gpurun --timeout 2400 -- 'bash tools/gpu_session.sh s1'
S=${1:-s1}
O=gpurun_out/$S
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s > $O/fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/summary.txt
for lib in product accinit; do
  if [ $lib = accinit ]; then export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_accinit.so; else unset DS_HIP_LIBRARY; fi
  timeout 300 python tests/hazard_probe.py poison >> $O/hazard.log 2>&1
  timeout 600 python tests/hazard_probe.py unet 60 >> $O/hazard.log 2>&1
  for k in 1 2 3; do timeout 300 python tests/hazard_probe.py pipe 4 >> $O/hazard.log 2>&1; done
done
unset DS_HIP_LIBRARY
grep '^{' $O/hazard.log | tee -a $O/summary.txt
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py > $O/gputest.log 2>&1; echo "gputest rc=$?" | tee -a $O/summary.txt
tail -3 $O/gputest.log | tee -a $O/summary.txt
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench rc=$?" | tee -a $O/summary.txt
head -c 600 $O/bench_cfg3.json | tee -a $O/summary.txt
grep -h "'test'" $O/fullsize.log | tee -a $O/summary.txt | tail -30
