#!/bin/bash
# round-3 experiment 17: a CONTINUOUS small CU share for the weight gradients instead of 108 CUs for 60 % of the time:
# ViT-B block cut into 2 (3) k-slices = 216 (324) equal work items walked by 72 / 81 / 108 workgroups
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"], r["config"]["final_loss"])'
run() { echo -n "cls $1: "; env $1 python bench.py $F 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
run "PM_GROUP_FORCE_SPLIT=0"
run "PM_GROUP_FORCE_SPLIT=2 PM_GROUP_BLOCKS_SLICED=72"
run "PM_GROUP_FORCE_SPLIT=2 PM_GROUP_BLOCKS_SLICED=80"
run "PM_GROUP_FORCE_SPLIT=2 PM_GROUP_BLOCKS_SLICED=108"
run "PM_GROUP_FORCE_SPLIT=3 PM_GROUP_BLOCKS_SLICED=81"
run "PM_GROUP_FORCE_SPLIT=3 PM_GROUP_BLOCKS_SLICED=64"
done
