#!/bin/bash
# gpurun with retries while every GPU slot of the pod is busy (status=transient, nothing charged):  tools/gpurun_retry.sh <timeout_s> '<command>'
T=$1; shift
for i in $(seq 1 20); do
  out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit 0
done
echo "$out"; exit 3
