#!/bin/bash
# rocprofv3 captures of a round: kernel stats (1 stream, default 2 streams, cfg5) and the step's HBM traffic (FETCH_SIZE / WRITE_SIZE passes)
S=${1:-final}; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s1 -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/prof_s1.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/prof_s2.log 2>&1
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $R/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof_cfg5.log 2>&1
timeout 1200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/stepF -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --streams 1 --graph 0 > $O/stepF.log 2>&1
timeout 1200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/stepW -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --streams 1 --graph 0 > $O/stepW.log 2>&1
cd $R
for c in s1 s2 cfg5; do f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$c.csv; find $O/prof_$c -name "*kernel_trace.csv" -delete; done
sf=$(find $O/stepF -name "*counter_collection.csv" | head -1); sw=$(find $O/stepW -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $sf $sw $O/pmc_hbm_traffic.json > $O/pmc_hbm_traffic.log 2>&1; echo "hbm summary rc=$?" | tee -a $O/summary.txt
find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*.db" -delete
du -sh $O | tee -a $O/summary.txt
