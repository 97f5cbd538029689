#!/bin/bash
# round 4, trip 7: long fp16 runs (400 steps: dynamic loss scale, skipped steps, finite losses, step-time drift), then the whole GPU suite once more
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="--precision fp16 --steps 400 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-c5 --no-mae"
for wl in "cls 64" "mae 256" "mae 64"; do
  set -- $wl
  timeout -k 10 400 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_t7_fp16_$1$2.json 2> gpurun_out/r4_t7_tmp.err || { tail -5 gpurun_out/r4_t7_tmp.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/r4_t7_fp16_$1$2.json')); print('fp16 400 steps $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms', d['step_ms'], d['loss_scaling'], 'loss', d['config']['final_loss'], d['config']['min_loss'])"
done | tee gpurun_out/r4_t7_fp16_long.txt
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4_t7_pytest.log 2>&1
RC=$?
tail -3 gpurun_out/r4_t7_pytest.log
exit $RC
