#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# rocprofv3 kernel stats of short one-stream cfg3 runs under two values of an environment switch: per-kernel ms per step side by side.
# usage: tools/gpu_prof_ab.sh <tag> <VAR> <value A> <value B>
S=$1; VAR=$2; A=$3; B=$4; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
for V in $A $B; do
  export $VAR=$V
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$V -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-mode 0 > $O/prof_$V.log 2>&1
  f=$(find $O/prof_$V -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$V.csv
  find $O/prof_$V -name "*kernel_trace.csv" -delete; find $O/prof_$V -name "*.db" -delete
done
cd $R
python3 - $O $A $B <<'PY' | tee $O/summary.txt
import csv,sys,re
O,A,B=sys.argv[1:4]
def load(v):
    d={}
    for r in csv.DictReader(open(f"{O}/kernel_stats_{v}.csv")):
        n=r["Name"]; n=re.sub(r"^_ZN12_GLOBAL__N_1\d+","",n); n=re.sub(r"^void \(anonymous namespace\)::","",n); n=re.sub(r"\(.*$","",n)[:70]
        d[n]=(int(r["Calls"]), float(r["TotalDurationNs"])/4e6)
    return d
a,b=load(A),load(B)
print(f"{'kernel':62s} {A:>14s} {B:>14s}   (calls, ms per step; 4 steps in trace)")
for k in sorted(set(a)|set(b), key=lambda k:-(a.get(k,(0,0))[1]+b.get(k,(0,0))[1]))[:40]:
    print(f"{k:62s} {a.get(k,(0,0))[0]:5d} {a.get(k,(0,0))[1]:8.2f} {b.get(k,(0,0))[0]:5d} {b.get(k,(0,0))[1]:8.2f}")
print("total", sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
PY
