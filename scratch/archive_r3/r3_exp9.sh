#!/bin/bash
# round-3 experiment 9: dgrad tile height vs the 108 CUs the grouped weight gradients hold (150 tiles of 256 rows + 108 = the chip)
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])'
run() { echo -n "cls PM_CFG_CLASS=$1 $2: "; env PM_CFG_CLASS=$1 $2 python bench.py $F 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
run 0,0,0,0,0
run 0,0,0,24,0
run 0,0,0,25,0
run 0,0,0,6,0
run 0,0,0,24,24
run 0,0,0,24,25
done
runm() { echo -n "mae PM_CFG_CLASS=$1 $2: "; env PM_CFG_CLASS=$1 $2 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
runm 0,0,0,0,0
runm 0,0,0,24,0
runm 0,0,0,25,0
done
