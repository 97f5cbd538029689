#!/bin/bash
# round 4, experiment 1: (a) grouped weight gradients stand-alone, 8-wave ping-pong body vs the 4-wave software-pipelined body
# (PM_GROUP_KERNEL=4); (b) same-box A/B of the live workload against the memorised single batch of rounds 1-3
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
echo "== 8-wave (shipped)"; timeout -k 10 200 python scratch/bench_wgroup.py || exit 1
echo "== 4-wave SWP"; PM_GROUP_KERNEL=4 timeout -k 10 200 python scratch/bench_wgroup.py || exit 1
echo "== correctness of the 4-wave body"; PM_GROUP_KERNEL=4 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "wgrad_group" 2>&1 | tail -3
} > gpurun_out/r4_exp1_wgroup.log 2>&1
cat gpurun_out/r4_exp1_wgroup.log
{
timeout -k 10 200 python scratch/bench_attn_layout.py || exit 1
PM_ATTN_HEADMAJOR=1 timeout -k 10 200 python scratch/bench_attn_layout.py || exit 1
} > gpurun_out/r4_exp1_attn_layout.log 2>&1
cat gpurun_out/r4_exp1_attn_layout.log
for rep in 1 2; do
for mode in live single; do
  flag=""; [ $mode = single ] && flag="--single-batch"
  timeout -k 10 300 python bench.py $flag --steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-mae > gpurun_out/r4_exp1_${mode}_${rep}.json 2>/dev/null || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/r4_exp1_${mode}_${rep}.json"))
c = d["config"]
print("${mode} ${rep}: cls full", d["value"], "loss", c["final_loss"], "| none", c.get("finetune_none_img_s"), c.get("finetune_none_final_loss"), "| head+1", c.get("finetune_head_plus_1_img_s"), c.get("finetune_head_plus_1_final_loss"), "| head+2", c.get("finetune_head_plus_2_img_s"), c.get("finetune_head_plus_2_final_loss"))
PY
done; done | tee gpurun_out/r4_exp1_live_vs_single.txt
