#!/bin/bash
# round-3 experiment 24: the final stream layout (main + aux0 [= second weight-gradient stream] + side, reserved before RCCL; default 4
# hardware queues) on every schedule: plain, forced world-1 RCCL, host input, host input + device augmentation, both together
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"])'
run() { echo -n "$*: "; python bench.py $F "$@" 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
run --workload cls
run --workload cls --force-sync
run --workload cls --input host
run --workload cls --input host --augment device
run --workload cls --input host --force-sync
run --workload cls --input host --augment device --force-sync
run --workload mae
run --workload mae --force-sync
done
