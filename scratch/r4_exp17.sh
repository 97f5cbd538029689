#!/bin/bash
# round 4, experiment 17: with the 128 x 128 kernel's batched epilogue, does the few-tiles rule want a higher cap on M (the half-batch
# proj / fc2 of the fine-tune forward, M = 6 304)?  + the full GPU suite and the default bench line on the final library
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
for rep in 1 2; do
for mm in 4096 8192 16384; do
  for wl in "cls 64" "mae 256"; do
    set -- $wl
    PM_FEW_TILES_MAXM=$mm timeout -k 10 200 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_exp17_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp17_tmp.json')); print('maxm $mm rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done; done | tee gpurun_out/r4_exp17_step.txt
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4_exp17_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r4_exp17_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py > gpurun_out/r4_t15_bench.json 2> gpurun_out/r4_t15_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_t15_bench.json"))
c = d["config"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
print({k: v for k, v in c.items() if k.endswith("_img_s")})
PY
