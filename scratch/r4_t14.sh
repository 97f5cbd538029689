#!/bin/bash
# round 4, trip 14: smoke() on the final library + the final default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 || exit 1
timeout -k 10 900 python bench.py > gpurun_out/r4_t14_bench.json 2> gpurun_out/r4_t14_bench.err; echo "bench rc $?"
tail -3 gpurun_out/r4_t14_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_t14_bench.json"))
c = d["config"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
print({k: v for k, v in c.items() if k.endswith("_img_s")})
PY
