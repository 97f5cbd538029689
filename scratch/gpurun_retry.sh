#!/bin/bash
# gpurun with re-tries while the pod has no free GPU slot (exit code 3: nothing ran, nothing was charged).  usage: gpurun_retry.sh TIMEOUT 'command'
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 150
done
exit 3
