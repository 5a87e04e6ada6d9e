#!/bin/bash
# rocprofv3 kernel stats of the t2v sphere loop at gen_pano_360's stage-1 geometry (tools/bench_sphere.py, 3 steps, eager launches so that
# every kernel is its own record)
O=$PWD/gpurun_out/${1:-sphereprof}; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o sphere -- python3 tools/bench_sphere.py --model t2v --steps 3 --graph 0 > $O/line.txt 2> $O/err.txt
echo "rc=$?"; tail -1 $O/line.txt | cut -c1-300
f=$(find $O/rp -name "*kernel_stats.csv" | head -1)
find $O/rp -name "*kernel_trace.csv" -delete; find $O/rp -name "*.db" -delete
[ -n "$f" ] && python tools/rocprof_step_summary.py "$f" 3 $O/summary.json > /dev/null && cp "$f" $O/kernel_stats.csv && python - <<PY
import json
j=json.load(open("$O/summary.json"))
print("kernel ms per step", round(j["kernel_ms_per_step"],1))
for k,v in sorted(j["families"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:8]: print(" ", k, round(v["ms_per_step"],1), round(v["share"],3), v["launches_per_step"])
PY
