#!/bin/bash
# usage: tools/gpu_retry.sh TIMEOUT 'command'  -- gpurun with retries while the pod's GPU slots are busy (exit 3)
t=$1; shift
for i in $(seq 1 30); do
  gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
