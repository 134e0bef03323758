#!/bin/bash
# retry wrapper around gpurun: waits while the pod's GPU slots are busy (exit code 3 = nothing charged)
# usage: tools/grun.sh <timeout_s> '<command>'
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
