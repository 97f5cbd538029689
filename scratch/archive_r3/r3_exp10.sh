#!/bin/bash
# round-3 experiment 10: CU budget of the grouped weight gradients beside 150-tile dgrads; wgrad split target for the MAE decoder
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])'
run() { echo -n "cls $1: "; env $1 python bench.py $F 2>/dev/null | python -c "$sel"; }
runm() { echo -n "mae $1: "; env $1 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
run PM_GROUP_BLOCKS=0
run PM_GROUP_BLOCKS=104
run PM_GROUP_BLOCKS=96
run PM_UNGROUP_TAIL=0
run PM_UNGROUP_TAIL=2
done
for i in 1 2; do
runm PM_GROUP_BLOCKS=0
runm PM_GROUP_SPLIT_TARGET=192
runm PM_GROUP_SPLIT_TARGET=160
runm PM_GROUP_SPLIT_TARGET=256
done
