#!/bin/bash
# round-3 experiment 23: three engine streams instead of four (the second weight-gradient stream = the second forward chain's stream),
# plain and with the forced world-1 RCCL schedule, 4 and 8 hardware queues
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"])'
run() { echo -n "merge=$1 queues=$2 $3 $4 $5: "; export PM_MERGE_AUX_SIDE2=$1 GPU_MAX_HW_QUEUES=$2; python bench.py $F $3 $4 $5 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
for m in 0 1; do
run $m 4 --workload cls
run $m 4 --workload cls --force-sync
run $m 8 --workload cls --force-sync
run $m 4 --workload mae
run $m 4 --workload mae --force-sync
run $m 8 --workload mae --force-sync
done; done
