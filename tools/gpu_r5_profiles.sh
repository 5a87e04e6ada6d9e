#!/bin/bash
# round 5: where a wide-mode evaluation spends its time (rocprofv3 kernel stats of tools/bench_wide.py), then the round profile set
O=gpurun_out/r5d; mkdir -p $O; R=$PWD
python tools/bench_wide.py 3 2>&1 | tail -2 | tee $O/wide_ms.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/wprof -- python3 $R/tools/bench_wide.py 2 > $R/$O/wprof.log 2>&1; echo "rc=$?"
cd $R
f=$(find $R/$O/wprof -name "*kernel_stats.csv" 2>/dev/null | head -1); cp $f $O/wide_kernel_stats.csv; head -25 $O/wide_kernel_stats.csv | cut -c1-200
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
bash tools/gpu_profiles.sh r5prof r5 > $O/profiles.log 2>&1; cat gpurun_out/r5prof/summary.txt
