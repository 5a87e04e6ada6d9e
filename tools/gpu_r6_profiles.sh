#!/bin/bash
# round 6 profile set on the final build.  DS_OPERAND_POLICY=f16: every profiled step runs in the model's own mode and no calibration
# evaluation enters the trace (the operand policy's strict steps are timed by the plain bench line, not profiled): the per-step accounting
# (warm-up + timed + one instrumented step) is the one the summary tools assume, and comparable with rounds 2-5.
export DS_OPERAND_POLICY=f16
bash tools/gpu_profiles.sh r6prof r6 > gpurun_out/r6prof.log 2>&1; cat gpurun_out/r6prof/summary.txt
bash tools/gpu_profile_rankshare.sh r6rank r6 > gpurun_out/r6rank.log 2>&1; cat gpurun_out/r6rank/summary.txt
# the rank share as round 5 measured it: col2 at tile batch 1 = two rank-steps of an 8-GPU cfg3 run; cond / uncond on two streams, and
# the [cond | uncond] pair batch on one stream
O=gpurun_out/r6rank
share() { tag=$1; shift; env "$@" timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 --wide-step 0 $SHARE_ARGS 2> $O/share_$tag.err | tail -1 > $O/share_$tag.json; echo "share $tag: $(grep -o '"ms_per_step": [0-9.]*' $O/share_$tag.json)" | tee -a $O/share_summary.txt; }
for rep in a b; do
  SHARE_ARGS="--streams 2" share split2_$rep DS_SPLIT_CFG=2
  SHARE_ARGS="--streams 1" share pair1s_$rep DS_SPLIT_CFG=0
done
