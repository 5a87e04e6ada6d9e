#!/bin/bash
# usage: tools/gpurun_lite.sh <timeout_s> <log> <command...>
# gpurun for PERFORMANCE calls: the golden archives (~180 MB of the ~195 MB snapshot; up to 165 s of metered upload per call) stay behind
# -- .gpurunignore is extended for the duration of the call and restored afterwards.  Test calls use tools/gpurun_retry.sh (full snapshot).
T=$1; LOG=$2; shift 2
cp .gpurunignore .gpurunignore.keep
trap 'mv -f .gpurunignore.keep .gpurunignore' EXIT
printf 'tests/golden/*.npz\n' >> .gpurunignore
for k in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1; rc=$?
  if [ $rc -ne 3 ]; then echo "gpurun rc=$rc (attempt $k)" >> $LOG; exit $rc; fi
  sleep 90
done
echo "gave up" >> $LOG; exit 3
