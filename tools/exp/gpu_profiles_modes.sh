#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# rocprofv3 kernel stats (one stream) of the two stricter precision modes: where their extra step time goes
S=${1:-profmodes}; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
for rd in f32outer f32; do
  export DS_RESIDUAL_DTYPE=$rd
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$rd -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/prof_$rd.log 2>&1; echo "stats $rd rc=$?" | tee -a $O/summary.txt
  f=$(find $O/prof_$rd -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$rd.csv; find $O/prof_$rd -name "*kernel_trace.csv" -delete; find $O/prof_$rd -name "*.db" -delete
  (cd $R && python3 tools/rocprof_step_summary.py $O/kernel_stats_$rd.csv 4 $O/r3_rocprof_step_summary_cfg3_1stream_$rd.json > /dev/null 2>&1); echo "summary $rd rc=$?" | tee -a $O/summary.txt
done
