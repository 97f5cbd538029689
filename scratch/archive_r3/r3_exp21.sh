#!/bin/bash
# round-3 experiment 21: the host-input / device-augmentation pipeline uses a fifth busy stream (DevicePrefetcher): 4 vs 8 hardware queues
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"])'
run() { echo -n "queues=${1:-default} $2 $3 $4 $5: "; if [ -n "$1" ]; then export GPU_MAX_HW_QUEUES=$1; else unset GPU_MAX_HW_QUEUES; fi; python bench.py $F $2 $3 $4 $5 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
run "" 
run "" --input host
run 8 --input host
run "" --input host --augment device
run 8 --input host --augment device
done
