#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}
# BASELINE config 5 (8192x1024x24f, 64 tiles/step, UNet at T=24) on the final build: rocprofv3 kernel stats of two steps and the
# whole-step HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, one step each, one stream, eager launches).
# usage: tools/gpu_profile_cfg5.sh <tag> <round prefix>
S=${1:-cfg5prof}; RP=${2:-r4}; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 > $O/prof.log 2>&1; echo "stats rc=$?" | tee -a $O/summary.txt
timeout 1500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/stepF -- python3 $R/bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 --streams 1 --graph 0 > $O/stepF.log 2>&1; echo "stepF rc=$?" | tee -a $O/summary.txt
timeout 1500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/stepW -- python3 $R/bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 --streams 1 --graph 0 > $O/stepW.log 2>&1; echo "stepW rc=$?" | tee -a $O/summary.txt
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_cfg5.csv; find $O/prof -name "*kernel_trace.csv" -delete
python3 tools/rocprof_step_summary.py $O/kernel_stats_cfg5.csv 4 $O/${RP}_rocprof_step_summary_cfg5.json > /dev/null 2>&1; echo "summary rc=$?" | tee -a $O/summary.txt
sf=$(find $O/stepF -name "*counter_collection.csv" | head -1); sw=$(find $O/stepW -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $sf $sw $O/${RP}_pmc_hbm_traffic_cfg5.json > $O/pmc_hbm_traffic.log 2>&1; echo "hbm summary rc=$?" | tee -a $O/summary.txt
find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*.db" -delete
grep -h -o '"ms_per_step": [0-9.]*' $O/prof.log | tee -a $O/summary.txt
