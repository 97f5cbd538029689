#!/bin/bash
# round 4, trip 3: fp16 parity at ViT-B, the PM_AUTO_VMCNT gate, the new default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fp16.py tests/test_gpu_schedule.py tests/test_gpu_parity_large.py -q -m gpu -s -k "fp16 or vmcnt or freeze" > gpurun_out/r4_t3_parity.log 2>&1
RC=$?
grep -E "\[parity\]|\[measured\]|passed|failed|Error|FAILED" gpurun_out/r4_t3_parity.log | tail -50
[ $RC -eq 0 ] || exit $RC
timeout -k 10 600 python bench.py > gpurun_out/r4_t3_bench.json 2> gpurun_out/r4_t3_bench.err
RC=$?
echo "bench rc $RC"; tail -5 gpurun_out/r4_t3_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_t3_bench.json"))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "dtype")}))
print(json.dumps(d["config"], indent=0))
print(json.dumps({k: v for k, v in d["roofline"].items() if not isinstance(v, (dict, list))}, indent=0))
PY
exit $RC
