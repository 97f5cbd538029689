#!/bin/bash
# round 4, experiment 11: forward / dgrad tiles of wide problems (N = 3 072, 2 304) walked in bands of tile columns (PM_TILE_BAND)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for band in 0 4 6; do
  echo "== PM_TILE_BAND=$band"
  PM_TILE_BAND=$band MS=6304,12608 timeout -k 10 200 python scratch/bench_gemm_smallm.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4_exp11_standalone.txt
PM_TILE_BAND=6 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm" 2>&1 | tail -2
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
for rep in 1 2; do
for band in 0 4 6; do
  for wl in "cls 64" "mae 256"; do
    set -- $wl
    PM_TILE_BAND=$band timeout -k 10 200 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_exp11_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp11_tmp.json')); print('band $band rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done; done | tee gpurun_out/r4_exp11_step.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for band in 0 6; do
PM_TILE_BAND=$band rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_r4_x11$band -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-torch-baseline --no-kernel-stats --no-mae --no-c5 --no-fp16 --no-fp32 --preheat 0.3 > $R/gpurun_out/r4_exp11_pmc_$band.log 2>&1
python3 - $R/gpurun_out/prof_r4_x11$band/f_counter_collection.csv $band <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "gemm_v3_kernel" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][34:100], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
for g, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"band {sys.argv[2]}: {g[0]} grid {g[1]}: {len(v)} launches, fetch {sum(v)/len(v)*2/1024:.1f} MB per launch")
PY
rm -rf $R/gpurun_out/prof_r4_x11$band
done | tee $R/gpurun_out/r4_exp11_pmc.txt
