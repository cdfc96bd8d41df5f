#!/bin/bash
# gpu_retry.sh <timeout-seconds> '<command>': gpurun with retries while no GPU slot / box is free (exit code 3 = nothing charged)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
