#!/bin/bash
# round 6, call K: the sphere stage and cfg5 on the final build (rocprofv3 kernel stats), own-mode steps only
export DS_OPERAND_POLICY=f16
bash tools/gpu_profile_sphere.sh r6sphere > gpurun_out/r6sphere.log 2>&1; tail -12 gpurun_out/r6sphere.log
bash tools/gpu_profile_cfg5.sh r6cfg5 r6 > gpurun_out/r6cfg5.log 2>&1; cat gpurun_out/r6cfg5/summary.txt
