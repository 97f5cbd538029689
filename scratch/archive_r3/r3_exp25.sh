#!/bin/bash
# round-3 experiment 25: data-parallel schedule + host-input pipeline (what a real training rank runs): the prefetcher's copies on a
# stream of their own (a fifth busy stream) vs on the engine's weight-gradient stream; 4 / 5 / 6 / 8 hardware queues
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40 --input host --augment device"
sel='import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"])'
run() { echo -n "prefetch_on_side=$1 queues=$2 $3: "; export PM_PREFETCH_ON_SIDE=$1 GPU_MAX_HW_QUEUES=$2; python bench.py $F $3 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
run 0 4
run 1 4
run 0 4 --force-sync
run 1 4 --force-sync
run 0 5 --force-sync
run 0 6 --force-sync
run 0 8 --force-sync
run 1 8 --force-sync
done
