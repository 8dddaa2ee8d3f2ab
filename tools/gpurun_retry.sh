#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'   - retries while the pool answers "transient" (no slot free; nothing charged)
T=$1; shift
for i in $(seq 1 40); do
  out=$(gpurun --timeout "$T" -- "$@" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 90; continue; fi
  echo "$out"
  exit 0
done
echo "gave up: no GPU slot"
exit 3
