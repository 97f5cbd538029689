#!/bin/bash
# round 4, experiment 20: fc1's GELU epilogue without the pre-activation store where no backward will read it (evaluation forward,
# linear probe, frozen blocks of the staged fine-tune)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_schedule.py -q -m gpu -x 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests/test_gpu_parity_large.py -q -m gpu -x -k "freeze or evaluate or classifier" 2>&1 | tail -2
timeout -k 10 900 python bench.py --no-mae --no-fp16 --no-fp32 --no-cpu-baseline --no-torch-baseline > gpurun_out/r4_exp20_bench.json 2> gpurun_out/r4_exp20_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_exp20_bench.json"))
c = d["config"]
print(d["value"], d["ms_per_step"])
print({k: v for k, v in c.items() if k.endswith("_img_s") or "parity_pass" in k})
PY
