#!/bin/bash
# round 4, trip 24: the caller's grad mode honoured inside the one-node functions (evaluation over a model with trainable blocks takes
# the forward-only path): whole GPU suite + default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4_t24_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r4_t24_pytest.log
[ $rc -eq 0 ] || { grep -E "Error|assert|FAILED" gpurun_out/r4_t24_pytest.log | head -20; exit $rc; }
timeout -k 10 900 python bench.py > gpurun_out/r4_t24_bench.json 2> gpurun_out/r4_t24_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_t24_bench.json"))
c = d["config"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_of_peak_at_held_clock"))
print({k: v for k, v in c.items() if k.endswith("_img_s")})
PY
