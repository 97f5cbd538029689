#!/bin/bash
# round-3 experiment 13 (MAE): CU budget of the k-sliced decoder weight gradients (192 work items hold 192 CUs for ~0.4 ms of every
# ~1.1 ms decoder block while the dgrad chain -- the critical path -- gets the other 64)
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])'
runm() { echo -n "mae $1: "; env $1 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
runm PM_GROUP_BLOCKS_SLICED=0
runm PM_GROUP_BLOCKS_SLICED=96
runm PM_GROUP_BLOCKS_SLICED=128
runm PM_GROUP_BLOCKS_SLICED=64
runm "PM_GROUP_BLOCKS_SLICED=96 PM_GROUP_SPLIT_TARGET=288"
done
