#!/bin/bash
# round 4, trip 25: smoke() and the forced world-1 RCCL schedule on the engine the round ends with (cls-row top block under GradSync)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_t25_smoke.txt 2>&1; rc=$?
tail -4 gpurun_out/r4_t25_smoke.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --force-sync --no-mae --no-fp32 --no-fp16 --no-cpu-baseline --no-torch-baseline --no-kernel-stats > gpurun_out/r4_t25_forcesync.json 2> gpurun_out/r4_t25_forcesync.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_t25_forcesync.json"))
c = d["config"]
print(d["value"], d["ms_per_step"], {k: v for k, v in c.items() if k.startswith("sync_") or k.endswith("_img_s") or k.endswith("parity_pass") or k == "world_size"})
PY
