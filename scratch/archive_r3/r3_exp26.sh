#!/bin/bash
# round-3 experiment 26: hipGraph replay of the whole step again, now that stream -> hardware-queue binding is understood: the graph's
# internal branch streams are created at instantiation (late) and may share the launch stream's queue.  4 / 6 / 8 queues, cls and MAE.
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"], r["host_enqueue_ms_per_step"], r["config"]["launch"])'
run() { echo -n "queues=$1 $2 $3 $4 $5: "; export GPU_MAX_HW_QUEUES=$1; python bench.py $F $2 $3 $4 $5 2>/dev/null | python -c "$sel"; }
for wl in cls mae; do
run 4 --workload $wl
run 4 --workload $wl --graph on
run 5 --workload $wl --graph on
run 6 --workload $wl --graph on
run 8 --workload $wl --graph on
done
