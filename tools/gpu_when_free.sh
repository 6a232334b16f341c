#!/bin/bash
# Retries a gpurun call ONLY while the pod has no free GPU slot (exit code 3: nothing ran, nothing was charged).
# usage: tools/gpu_when_free.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 30); do
    gpurun --timeout "$t" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 150
done
exit 3
