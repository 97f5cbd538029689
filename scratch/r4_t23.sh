#!/bin/bash
# round 4, trip 23: the library and engine the round ends with (cls-row top block): whole GPU suite, default bench line, cls profile set
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4_t23_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r4_t23_pytest.log
[ $rc -eq 0 ] || { grep -E "Error|assert|FAILED" gpurun_out/r4_t23_pytest.log | head -20; exit $rc; }
timeout -k 10 900 python bench.py > gpurun_out/r4_t23_bench.json 2> gpurun_out/r4_t23_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_t23_bench.json"))
c = d["config"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_of_peak_at_held_clock"))
print({k: v for k, v in c.items() if k.endswith("_img_s") or k.endswith("parity_pass")})
PY
bash scratch/profile_round4.sh r4j_cls r4_j_cls_bs64 > gpurun_out/r4_prof_clsj.log 2>&1 || { tail -20 gpurun_out/r4_prof_clsj.log; exit 1; }
ls gpurun_out/profiles_r4 | grep r4_j
