#!/bin/bash
# is the MAE bench's final loss reproducible run to run?  (exp 18 saw 1.12 once against 1.10534)
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40 --workload mae"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["config"]["final_loss"])'
for i in 1 2 3 4 5; do for v in 0 1; do echo -n "PM_TWO_GROUPS=$v: "; PM_TWO_GROUPS=$v python bench.py $F 2>/dev/null | python -c "$sel"; done; done
