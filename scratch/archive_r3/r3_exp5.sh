#!/bin/bash
# round-3 experiment 5: forward split factor (independent sub-batch chains) for both workloads
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])'
for sp in 2 3 4; do
echo "== cls split=$sp =="; PM_SPLIT_FWD=$sp python bench.py $F 2>/dev/null | python -c "$sel"
echo "== MAE split=$sp =="; PM_SPLIT_FWD=$sp python bench.py --workload mae $F 2>/dev/null | python -c "$sel"
done
echo "== MAE split=2 overlap adamw =="; PM_OVERLAP_ADAMW=1 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"
