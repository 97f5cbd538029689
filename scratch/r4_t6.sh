#!/bin/bash
# round 4, trip 6: fp16 training loops + checkpoints test, the default bench line with the fp32-mode record
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fp16.py -q -m gpu -x -s 2>&1 | grep -E "measured|passed|failed|Error|assert" | tail -12
timeout -k 10 500 python bench.py > gpurun_out/r4_t6_bench.json 2> gpurun_out/r4_t6_bench.err
RC=$?
echo "bench rc $RC"; tail -3 gpurun_out/r4_t6_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_t6_bench.json"))
c = d["config"]
print(d["value"], {k: v for k, v in c.items() if k.startswith("cls_fp32") or k.startswith("cls_fp16_img") or k.startswith("mae_bs64_img")})
PY
exit $RC
