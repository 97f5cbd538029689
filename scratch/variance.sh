#!/bin/bash
# usage: bash scratch/variance.sh <outdir> [n]  -- the cls bench line n times in a row with and without the parity block, per-step spread
O=$1; N=${2:-4}; mkdir -p $O
for i in $(seq 1 $N); do
  python bench.py --no-cpu-baseline --no-torch-baseline --no-mae --no-kernel-stats > $O/parity_$i.json 2> $O/parity_$i.err
  python bench.py --no-cpu-baseline --no-torch-baseline --no-mae --no-kernel-stats --no-parity > $O/bare_$i.json 2> $O/bare_$i.err
done
python - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["step_ms"])
    except Exception as e:
        print(f, "FAILED", e)
PY
