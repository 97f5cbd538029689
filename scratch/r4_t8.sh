#!/bin/bash
# round 4, trip 8: run-to-run spread of the headline on one box (five short runs), 400-step bf16 runs (drift), MAE bs=64 with the host busy
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="--no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
{
for i in 1 2 3 4 5; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 10 $B > gpurun_out/r4_t8_tmp.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r4_t8_tmp.json')); print('cls bs64 bf16 run $i:', d['value'], 'img/s', d['ms_per_step'], 'ms', d['step_ms'])"
done
for wl in "cls 64" "mae 256" "mae 64"; do
  set -- $wl
  timeout -k 10 300 python bench.py --workload $1 --batch $2 --steps 400 --warmup 10 $B > gpurun_out/r4_t8_tmp.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r4_t8_tmp.json')); print('bf16 400 steps $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms', d['step_ms'], 'loss', d['config']['final_loss'], d['config']['min_loss'])"
done
timeout -k 10 300 python bench.py --workload mae --batch 64 --steps 60 --warmup 10 --host-busy 16 $B > gpurun_out/r4_t8_tmp.json 2>/dev/null || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r4_t8_tmp.json')); print('mae bs64, 16 spinning processes beside it:', d['value'], 'img/s alone', d['busy_host'])"
} | tee gpurun_out/r4_t8_spread_and_drift.txt
