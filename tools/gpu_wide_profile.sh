#!/bin/bash
# where a wide-mode evaluation spends its time: rocprofv3 kernel stats of tools/bench_wide.py
O=gpurun_out/${1:-wideprof}; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/wprof -- python3 $R/tools/bench_wide.py 2 > $R/$O/wprof.log 2>&1; echo "rc=$?"
cd $R
f=$(find $R/$O/wprof -name "*kernel_stats.csv" 2>/dev/null | head -1); cp $f $O/wide_kernel_stats.csv; head -12 $O/wide_kernel_stats.csv | cut -c1-160
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; tail -2 $O/wprof.log
