#!/usr/bin/env bash
# gpurun with retries while the pod's GPU slots are busy (exit 3 = nothing charged).  usage: gpurun_retry.sh TIMEOUT 'command'
t="$1"; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [[ $rc != 3 ]]; then exit $rc; fi
  sleep 90
done
exit 3
