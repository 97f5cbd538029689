#!/bin/bash
# round 4, experiment 10: dh = 32 attention (MAE decoder): the two heads that share 128-B lines mapped to the same XCD (default) against
# the identity map (libpolypmae_nopair.so, -DPM_ATTN_NO_PAIR_MAP): stand-alone, PMC fetch, MAE step
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
NOPAIR=$PWD/ssl4polyp_amd/lib/libpolypmae_nopair.so
{
echo "== identity map"; POLYPMAE_LIB=$NOPAIR timeout -k 10 200 python scratch/bench_attn.py 2>&1 | grep -v amdgpu.ids
echo "== pair map";     timeout -k 10 200 python scratch/bench_attn.py 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r4_exp10_standalone.txt
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k attention 2>&1 | tail -2
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
for rep in 1 2; do
for lib in nopair pair; do
  for wl in "mae 256" "mae 64"; do
    set -- $wl
    L=""; [ $lib = nopair ] && L=$NOPAIR
    POLYPMAE_LIB=$L timeout -k 10 200 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_exp10_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp10_tmp.json')); print('$lib rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done; done | tee gpurun_out/r4_exp10_step.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in nopair pair; do
L=""; [ $lib = nopair ] && L=$NOPAIR
POLYPMAE_LIB=$L rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_r4_x10$lib -o f --output-format csv -- python3 $R/bench.py --workload mae --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-torch-baseline --no-kernel-stats --preheat 0.3 > $R/gpurun_out/r4_exp10_pmc_$lib.log 2>&1
python3 - $R/gpurun_out/prof_r4_x10$lib/f_counter_collection.csv $lib <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "attn_" in r["Kernel_Name"] and "Li32E" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][20:48], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
for g, v in sorted(agg.items()):
    print(f"{sys.argv[2]}: {g}: {len(v)} launches, fetch {sum(v)/len(v)*2/1024:.1f} MB per launch (x2 corrected)")
PY
rm -rf $R/gpurun_out/prof_r4_x10$lib
done | tee $R/gpurun_out/r4_exp10_pmc.txt
