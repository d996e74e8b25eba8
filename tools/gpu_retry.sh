#!/bin/bash
# gpurun with retries while no slot is free (exit 3): tools/gpu_retry.sh <timeout-s> '<command>'   (log -> stdout)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
