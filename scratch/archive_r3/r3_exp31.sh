#!/bin/bash
# round-3 experiment 31: AdamW beside the next forward (PM_OVERLAP_ADAMW) re-checked under the final stream layout
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"])'
run() { echo -n "PM_OVERLAP_ADAMW=$1 $2: "; PM_OVERLAP_ADAMW=$1 python bench.py $F --workload $2 2>/dev/null | python -c "$sel"; }
for i in 1 2 3; do run 1 cls; run 0 cls; run 0 mae; run 1 mae; done
