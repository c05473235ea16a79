#!/usr/bin/env bash
# gpurun with retries while the pod has no free GPU slot (exit code 3: nothing charged).
#   usage: scripts/gpurun_retry.sh <timeout-seconds> '<command>'
T="$1"; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
