#!/bin/bash
# round-3 experiment 28: the lowest block's weight gradients (nothing but the embedding's backward follows): per-GEMM split-K launches
# on all CUs (PM_UNGROUP_TAIL=1, the default since round 2) vs the grouped two-launch schedule like every other block (=0)
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"])'
run() { echo -n "PM_UNGROUP_TAIL=$1 $2: "; PM_UNGROUP_TAIL=$1 python bench.py $F --workload $2 2>/dev/null | python -c "$sel"; }
for i in 1 2 3; do run 1 cls; run 0 cls; run 1 mae; run 0 mae; done
