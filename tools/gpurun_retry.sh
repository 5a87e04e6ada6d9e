#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> <log> <command...>   -- retries `gpurun` while no box / slot is free (exit code 3)
T=$1; LOG=$2; shift 2
for k in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1; rc=$?
  if [ $rc -ne 3 ]; then echo "gpurun rc=$rc (attempt $k)" >> $LOG; exit $rc; fi
  sleep 90
done
echo "gave up" >> $LOG; exit 3
