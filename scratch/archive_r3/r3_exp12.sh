#!/bin/bash
# round-3 experiment 12: CU budget of the remaining split-K weight gradients (tail block, embeddings) now that they run at the very end
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])'
run() { echo -n "cls $1: "; env $1 python bench.py $F 2>/dev/null | python -c "$sel"; }
runm() { echo -n "mae $1: "; env $1 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"; }
for i in 1 2; do
run PM_WGRAD_BLOCKS=128
run PM_WGRAD_BLOCKS=192
run PM_WGRAD_BLOCKS=256
done
for i in 1 2; do
runm PM_WGRAD_BLOCKS=128
runm PM_WGRAD_BLOCKS=256
done
