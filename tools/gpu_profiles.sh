#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# (a bench.py run = warm-up + timed steps + ONE per-rank instrumented step: 1 + 3 + 1 = 5 steps in each stats trace)
# Round profile set on the final build: rocprofv3 kernel stats (1 stream = per-launch durations comparable with bench.py's HIP-event
# figure; default 2 streams), per-family step summaries, whole-step HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes), and the
# SQ / LDS counter table of the top shapes.  Usage: tools/gpu_profiles.sh <tag> <round-prefix, e.g. r3>
S=${1:-prof}; RP=${2:-r3}; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s1 -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 > $O/prof_s1.log 2>&1; echo "stats 1 stream rc=$?" | tee -a $O/summary.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 > $O/prof_s2.log 2>&1; echo "stats 2 streams rc=$?" | tee -a $O/summary.txt
timeout 1200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/stepF -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 --streams 1 --graph 0 > $O/stepF.log 2>&1; echo "stepF rc=$?" | tee -a $O/summary.txt
timeout 1200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/stepW -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 --streams 1 --graph 0 > $O/stepW.log 2>&1; echo "stepW rc=$?" | tee -a $O/summary.txt
cd $R
for c in s1 s2; do f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$c.csv; find $O/prof_$c -name "*kernel_trace.csv" -delete; done
python3 tools/rocprof_step_summary.py $O/kernel_stats_s1.csv 5 $O/${RP}_rocprof_step_summary_cfg3_1stream.json $O/prof_s1.log > /dev/null 2>&1; echo "summary s1 rc=$?" | tee -a $O/summary.txt
python3 tools/rocprof_step_summary.py $O/kernel_stats_s2.csv 5 $O/${RP}_rocprof_step_summary_cfg3.json $O/prof_s2.log > /dev/null 2>&1; echo "summary s2 rc=$?" | tee -a $O/summary.txt
sf=$(find $O/stepF -name "*counter_collection.csv" | head -1); sw=$(find $O/stepW -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $sf $sw $O/${RP}_pmc_hbm_traffic.json $O/stepF.log > $O/pmc_hbm_traffic.log 2>&1; echo "hbm summary rc=$?" | tee -a $O/summary.txt
bash tools/gpu_pmc_shapes.sh $S/shapes > /dev/null 2>&1; echo "pmc shapes rc=$?" | tee -a $O/summary.txt
find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*.db" -delete
du -sh $O | tee -a $O/summary.txt
