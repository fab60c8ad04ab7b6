#!/bin/bash
# gpurun with retries on "no box / slot free right now" (exit code 3: nothing was charged).  Any other outcome is final.
# usage: tools/gpurun_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
