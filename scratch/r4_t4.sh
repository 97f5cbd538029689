#!/bin/bash
# round 4, trip 4: the whole GPU suite + smoke, then the N > 1 code path rehearsed on one GPU (2 gloo ranks on cuda:0) and the forced world-1 RCCL schedule
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4_t4_pytest.log 2>&1
RC=$?
tail -4 gpurun_out/r4_t4_pytest.log
[ $RC -eq 0 ] || { grep -E "Error|assert|FAILED" gpurun_out/r4_t4_pytest.log | head -20; exit $RC; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_t4_smoke.log || exit 1
B="--steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-parity --no-mae --no-c5 --no-fp16"
BENCH_REHEARSAL=1 timeout -k 10 400 python bench.py --gpus 2 $B > gpurun_out/r4_t4_rehearsal2.json 2> gpurun_out/r4_t4_rehearsal2.err || { tail -20 gpurun_out/r4_t4_rehearsal2.err; exit 1; }
timeout -k 10 300 python bench.py --force-sync $B > gpurun_out/r4_t4_forcesync.json 2> gpurun_out/r4_t4_forcesync.err || { tail -20 gpurun_out/r4_t4_forcesync.err; exit 1; }
python - <<'PY'
import json
for n in ("rehearsal2", "forcesync"):
    d = json.load(open(f"gpurun_out/r4_t4_{n}.json"))
    c = d["config"]
    print(n, d["value"], d["n_gpus"], {k: c[k] for k in c if k.startswith("sync_") or k in ("world_size", "backend", "ranks_seen", "local_ranks", "devices", "distinct_local_devices", "parallelism")})
PY
