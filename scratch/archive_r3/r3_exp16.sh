#!/bin/bash
# round-3 experiment 16: the main chain on a high-priority stream
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])'
run() { echo -n "cls $1: "; env $1 python bench.py $F 2>/dev/null | python -c "$sel"; }
runm() { echo -n "mae $1: "; env $1 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
run PM_MAIN_PRIO=0
run PM_MAIN_PRIO=-1
done
for i in 1 2; do
runm PM_MAIN_PRIO=0
runm PM_MAIN_PRIO=-1
done
