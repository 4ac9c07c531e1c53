#!/bin/bash
# usage: gpurun_retry.sh <logfile> <timeout_s> '<command>'   -- retries while the pod has no free GPU slot (exit code 3)
log=$1; to=$2; shift 2
for attempt in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" $log; then break; fi
  sleep 60
done
echo "[gpurun_retry] done rc=$rc attempts=$attempt" >> $log
