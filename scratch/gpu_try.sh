#!/bin/bash
# gpu_try.sh <timeout> <command...>: gpurun, retried only while no GPU slot is free (exit 3: nothing ran, nothing charged)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gpu_try.out 2>&1; rc=$?
  if grep -q "status=transient" /tmp/gpu_try.out; then sleep 90; continue; fi
  break
done
cat /tmp/gpu_try.out
exit $rc
