#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# rocprofv3 kernel stats of an 8-GPU rank's share of cfg3 emulated on one GPU (col2 = two columns = twice the share; one
# tile per batch, one stream): sum of kernel time against the wall time of the step -> what part of the share is launch gaps
S=${1:-rankshare}; RP=${2:-r4}; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/bench.py --config col2 --steps 6 --warmup 2 --streams 1 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/wall.json 2> $O/wall.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --config col2 --steps 6 --warmup 2 --streams 1 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/prof.json 2> $O/prof.err; echo "stats rc=$?" | tee -a $O/summary.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv; find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete
(cd $R && python3 tools/rocprof_step_summary.py $O/kernel_stats.csv 9 $O/${RP}_rocprof_step_summary_rankshare_tb1.json > /dev/null 2>&1); echo "summary rc=$?" | tee -a $O/summary.txt
grep -h -o '"ms_per_step": [0-9.]*' $O/wall.json $O/prof.json | tee -a $O/summary.txt
