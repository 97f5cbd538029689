#!/bin/bash
# round 4, experiment 13: dispatch rules tuned on ViT-B (few-tiles rule, two forward chains) at the shapes of ViT-L / ViT-H (bs = 64)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for model in mae_vit_huge_patch14 mae_vit_large_patch16; do
for cfg in "128 2" "0 2" "128 1" "0 1"; do
  set -- $cfg
  echo -n "PM_FEW_TILES=$1 PM_SPLIT_FWD=$2: "
  PM_FEW_TILES=$1 PM_SPLIT_FWD=$2 timeout -k 10 200 python scratch/bench_huge.py --model $model --batch 64 --precision bf16 2>&1 | grep "ms/step" || exit 1
done; done | tee gpurun_out/r4_exp13_factories_dispatch.txt
timeout -k 10 900 python bench.py > gpurun_out/r4_t12_bench.json 2> gpurun_out/r4_t12_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_t12_bench.json"))
print(d["value"], d["ms_per_step"], {k: v for k, v in d["config"].items() if "vith" in k})
PY
