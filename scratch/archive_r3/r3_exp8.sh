#!/bin/bash
# round-3 experiment 8: ring-kernel variant per GEMM class, IN-STEP (stand-alone timings do not predict the step: the dGELU dgrad
# on cfg 8 is 1.7 us slower alone and +0.6 % in the step).  PM_CFG_CLASS = nt_store, nt_gelu, nt_residual, nn_store, nn_dgelu
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])'
run() { echo -n "cls PM_CFG_CLASS=$1: "; PM_CFG_CLASS=$1 python bench.py $F 2>/dev/null | python -c "$sel"; }
run 0,0,0,0,0
run 24,0,0,0,0
run 0,24,0,0,0
run 0,0,24,0,0
run 0,0,8,0,0
run 0,0,9,0,0
run 0,0,0,24,0
run 0,0,0,8,0
run 0,0,0,0,24
run 0,0,0,0,25
run 0,0,0,0,0
runm() { echo -n "mae PM_CFG_CLASS=$1: "; PM_CFG_CLASS=$1 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"; }
runm 0,0,0,0,0
runm 24,0,0,0,0
runm 0,24,0,0,0
runm 0,0,24,0,0
runm 0,0,0,24,0
runm 0,0,0,0,24
runm 0,0,0,0,0
