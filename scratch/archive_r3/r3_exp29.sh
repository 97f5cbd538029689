#!/bin/bash
# round-3 experiment 29: the k-sliced decoder launch under the final stream layout: slices (PM_GROUP_SPLIT_TARGET 96 / 144 / 224 =
# 2 / 3 / 4 slices of the 48 tiles) and the CU cap (PM_GROUP_BLOCKS_SLICED)
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40 --workload mae"
sel='import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"])'
run() { echo -n "target=$1 cap=$2: "; PM_GROUP_SPLIT_TARGET=$1 PM_GROUP_BLOCKS_SLICED=$2 python bench.py $F 2>/dev/null | python -c "$sel"; }
for i in 1 2; do run 224 0; run 144 0; run 96 0; run 224 128; run 224 96; run 288 0; done
