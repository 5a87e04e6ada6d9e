#!/bin/bash
export DS_BENCH_OTHER_MODE=${DS_BENCH_OTHER_MODE:-0}   # the A/B and sweep tools time ONE mode per bench.py run
# the 8-GPU layout of cfg3 in small: col2 (two columns) on two ranks = one column per rank, one tile per level and rank (cond / uncond
# on two streams, components mode, one all-gather per step), both ranks on this GPU through gloo; digest against the one-process run
O=gpurun_out/${1:-rehearse}; mkdir -p $O
export PYTHONUNBUFFERED=1 GLOO_SOCKET_IFNAME=lo
timeout 900 python bench.py --config col2 --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/one.json 2> $O/one.err
DS_DIST_BACKEND=gloo DS_BENCH_DEVICE=0 timeout 1200 python bench.py --config col2 --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/two.json 2> $O/two.err
for f in $O/one.json $O/two.json; do echo "$(basename $f) $(grep -o '"n_gpus": [0-9]*' $f) $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"result_sha256": {[^}]*}' $f)"; done | tee $O/summary.txt
