#!/bin/bash
# round 4, trip 13: full GPU suite on the ViT-H-capable library + the ViT-H sub-record under the forced world-1 RCCL schedule
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4_t13_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r4_t13_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --force-sync --no-fp32 --no-c5 --no-fp16 --no-cpu-baseline --no-torch-baseline --no-kernel-stats > gpurun_out/r4_t13_forcesync.json 2> gpurun_out/r4_t13_forcesync.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_t13_forcesync.json"))
c = d["config"]
print(d["value"], d["ms_per_step"], {k: v for k, v in c.items() if "vith" in k or k.startswith("sync_") or k in ("world_size", "collectives_per_step")})
PY
