#!/bin/bash
# round 4, experiment 7: the 36-tile (proj, qkv) weight-gradient launch confined to 4 (or 2) of the 8 XCDs: longer runs of tiles per L2
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -x -k "wgrad or block_backward" 2>&1 | tail -2
PM_GROUP_CONFINE_WORK=40 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -x -k "wgrad or block_backward" 2>&1 | tail -2
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-c5 --no-mae"
for rep in 1 2; do
for conf in "0 4" "40 4" "40 2" "80 4"; do
  set -- $conf
  for wl in "cls 64" "mae 256"; do
    PM_GROUP_CONFINE_WORK=$1 PM_GROUP_CONFINE_XCDS=$2 timeout -k 10 200 python bench.py --workload ${wl% *} --batch ${wl#* } $B > gpurun_out/r4_exp7_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp7_tmp.json')); print('confine work<=$1 xcds $2 rep $rep $wl:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done; done | tee gpurun_out/r4_exp7_confine.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in 0 40; do
PM_GROUP_CONFINE_WORK=$w rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_r4_conf$w -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-torch-baseline --no-mae --no-c5 --no-fp16 --no-kernel-stats --preheat 0.3 > $R/gpurun_out/r4_exp7_pmc$w.log 2>&1
python3 - $R/gpurun_out/prof_r4_conf$w/f_counter_collection.csv $w <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "wgrad_group_kernel" in r["Kernel_Name"]:
        agg[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
for g, v in sorted(agg.items()):
    print(f"confine_work {sys.argv[2]}: wgrad_group grid {g} threads: {len(v)} launches, fetch {sum(v)/len(v)*2/1024:.1f} MB per launch (x2 corrected)")
PY
rm -rf $R/gpurun_out/prof_r4_conf$w
done | tee $R/gpurun_out/r4_exp7_confine_pmc.txt
