#!/bin/bash
# round 4, trip 17: the N > 1 code path of the final bench (2 gloo ranks on the one GPU), incl. the ViT-H sub-record under DataParallel
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
BENCH_REHEARSAL=1 timeout -k 10 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-fp32 --no-c5 > gpurun_out/r4_t17_rehearsal.json 2> gpurun_out/r4_t17_rehearsal.err; echo "bench rc $?"
tail -3 gpurun_out/r4_t17_rehearsal.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_t17_rehearsal.json"))
c = d["config"]
print(d["n_gpus"], d["value"], d["ms_per_step"])
print({k: v for k, v in c.items() if k.endswith("_img_s") or k.startswith("sync_") or k in ("world_size", "ranks_seen", "devices", "backend")})
PY
