#!/bin/bash
# usage: bash scratch/variance_full.sh <outdir> [n]  -- the default bench line (minus the CPU / torch baselines) n times in a row
O=$1; N=${2:-5}; mkdir -p $O
for i in $(seq 1 $N); do
  python bench.py --no-cpu-baseline --no-torch-baseline > $O/full_$i.json 2> $O/full_$i.err
done
python - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["step_ms"], "mae", d["mae"]["value"], d["mae"]["step_ms"])
    except Exception as e:
        print(f, "FAILED", e)
PY
