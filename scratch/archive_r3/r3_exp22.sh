#!/bin/bash
# round-3 experiment 22: data-parallel schedule (forced world-1 RCCL) TOGETHER with the host-input pipeline: 4 / 8 / 16 hardware queues
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40 --force-sync"
sel='import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"])'
run() { echo -n "queues=$1 $2 $3 $4 $5: "; export GPU_MAX_HW_QUEUES=$1; python bench.py $F $2 $3 $4 $5 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
for q in 4 6 8 16; do
run $q --input host
run $q --input host --augment device
done; done
