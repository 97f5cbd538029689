#!/bin/bash
# round-3 experiment 15: the 16x16x32 forward kernel (cfg 40) in the step, per GEMM class
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])'
run() { echo -n "cls PM_CFG_CLASS=$1: "; PM_CFG_CLASS=$1 python bench.py $F 2>/dev/null | python -c "$sel"; }
runm() { echo -n "mae PM_CFG_CLASS=$1: "; PM_CFG_CLASS=$1 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
run 0,0,0,0,0
run 40,0,0,0,0
run 0,40,0,0,0
run 40,40,0,0,0
run 40,40,40,0,0
done
for i in 1 2; do
runm 0,0,0,0,0
runm 40,40,0,0,0
runm 40,40,40,0,0
done
