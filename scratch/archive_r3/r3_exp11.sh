#!/bin/bash
# round-3 experiment 11: embedding backward before the side-stream join (PM_DEFER_JOIN)
python -m pytest tests/test_gpu_schedule.py tests/test_gpu_models.py -m gpu -x -q 2>&1 | tail -3
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])'
run() { echo -n "cls $1: "; env $1 python bench.py $F 2>/dev/null | python -c "$sel"; }
runm() { echo -n "mae $1: "; env $1 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"; }
for i in 1 2 3; do
run PM_DEFER_JOIN=0
run PM_DEFER_JOIN=1
done
for i in 1 2; do
runm PM_DEFER_JOIN=0
runm PM_DEFER_JOIN=1
done
