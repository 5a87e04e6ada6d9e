#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# the kernel-stats half of gpu_profiles.sh (after a change that does not touch csrc/gemm.hip: the PMC passes stay valid)
S=${1:-profstats}; RP=${2:-r3}; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s1 -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 > $O/prof_s1.log 2>&1; echo "stats 1 stream rc=$?" | tee -a $O/summary.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 > $O/prof_s2.log 2>&1; echo "stats 2 streams rc=$?" | tee -a $O/summary.txt
cd $R
for c in s1 s2; do f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$c.csv; find $O/prof_$c -name "*kernel_trace.csv" -delete; find $O/prof_$c -name "*.db" -delete; done
python3 tools/rocprof_step_summary.py $O/kernel_stats_s1.csv 5 $O/${RP}_rocprof_step_summary_cfg3_1stream.json $O/prof_s1.log > /dev/null 2>&1; echo "summary s1 rc=$?" | tee -a $O/summary.txt
python3 tools/rocprof_step_summary.py $O/kernel_stats_s2.csv 5 $O/${RP}_rocprof_step_summary_cfg3.json $O/prof_s2.log > /dev/null 2>&1; echo "summary s2 rc=$?" | tee -a $O/summary.txt
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
