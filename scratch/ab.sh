#!/bin/bash
# usage: bash scratch/ab.sh <outdir> "<name>:<ENV=.. ENV=..>" ...   -- bench.py (cls + mae sub-record) per variant, one line each
O=$1; shift; mkdir -p $O
B="--no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats"
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs python bench.py $B > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    m = d.get("mae", {})
    print(f"{sys.argv[2]:12s} cls {d['value']:8.1f} img/s {d['ms_per_step']:7.3f} ms (host {d['host_enqueue_ms_per_step']:.2f})   mae {m.get('value', 0):8.1f} img/s {m.get('ms_per_step', 0):7.3f} ms (host {m.get('host_enqueue_ms_per_step', 0):.2f})")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
