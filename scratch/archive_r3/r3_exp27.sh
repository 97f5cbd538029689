#!/bin/bash
# round-3 experiment 27: the k-sliced MAE decoder block (48 tiles x 3 slices + reduce, ~144 CUs for 0.4 ms of a 1.1-ms block) as two
# WHOLE-K launches on the two side streams instead (needs the experimental two_groups modes 2 / 3 of the launcher, not kept): mode 2 = 256x256 tiles (32 + 16 CUs for ~1.3 ms), mode 3 = 256x128 tiles
# (64 + 32 CUs for ~0.7 ms); no slabs, no reduce
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40 --workload mae"
sel='import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"], r["config"]["final_loss"])'
run() { echo -n "PM_TWO_GROUPS_SLICED=$1 mode=$2: "; PM_TWO_GROUPS_SLICED=$1 PM_TWO_GROUPS_SLICED_MODE=$2 python bench.py $F 2>/dev/null | python -c "$sel"; }
for i in 1 2 3; do run 0 0; run 1 3; done
