#!/bin/bash
O=gpurun_out/dbg_mr; mkdir -p $O
export PYTHONUNBUFFERED=1 GLOO_SOCKET_IFNAME=lo
run2() { d=$(mktemp -d); python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) tests/multirank_worker.py $d grid4x2 reference 2>&1 | grep STEP_HASHES; }
run1() { d=$(mktemp -d); python tests/multirank_worker.py $d grid4x2 reference 2>&1 | grep STEP_HASHES; }
echo "== single (graph, 2 streams)" | tee -a $O/log.txt; run1 | tee -a $O/log.txt
echo "== single eager" | tee -a $O/log.txt; DS_WORKER_EAGER=1 run1 | tee -a $O/log.txt
echo "== 2 ranks auto" | tee -a $O/log.txt; run2 | tee -a $O/log.txt
echo "== 2 ranks levels" | tee -a $O/log.txt; DS_SHARE_MODE=levels run2 | tee -a $O/log.txt
echo "== 2 ranks auto eager" | tee -a $O/log.txt; DS_WORKER_EAGER=1 run2 | tee -a $O/log.txt
echo "== 2 ranks levels eager" | tee -a $O/log.txt; DS_WORKER_EAGER=1 DS_SHARE_MODE=levels run2 | tee -a $O/log.txt
