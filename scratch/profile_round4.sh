#!/bin/bash
# usage (GPU box, repo root): bash scratch/profile_round4.sh <tag> <name> [bench args...]
# kernel-trace stats + timeline + FETCH/WRITE PMC passes + MFMA-utilisation PMC pass of bench.py (one workload); the summaries are
# written to gpurun_out/profiles_r4/<name>_* (what gets committed under profiles/), the raw traces are deleted on the box
# (gpurun merges at most 64 MiB back).
TAG=$1; NAME=$2; shift; shift
REPO=$PWD; OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT $REPO/gpurun_out/profiles_r4
B="--no-cpu-baseline --no-parity --no-torch-baseline --no-mae --no-c5 --no-fp16 --no-fp32 --preheat 0.3"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 3 $B "$@" > $OUT/stats.log 2>&1
grep "^{\"metric\"" $OUT/stats.log > $OUT/bench.json
rocprofv3 --kernel-trace -d $OUT/trace -o t -- python3 $REPO/bench.py --steps 6 --warmup 3 $B --no-kernel-stats "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 2 $B --no-kernel-stats "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 2 $B --no-kernel-stats "$@" > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/mfma -o m --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 2 $B --no-kernel-stats "$@" > $OUT/mfma.log 2>&1
cd $REPO
python3 scratch/trace_timeline.py $OUT/trace/t_results.db -2 > $OUT/timeline.txt 2>&1
python3 scratch/pmc_mfma_post.py $OUT/mfma/m_counter_collection.csv > $OUT/mfma_util.txt 2>&1
mkdir -p profiles_tmp && python3 - "$TAG" "$NAME" <<'PY'
import os, sys, re
src = open("scratch/profile_post2.py").read().replace('f"profiles/', 'f"gpurun_out/profiles_r4/')
sys.argv = ["profile_post2.py", sys.argv[1], sys.argv[2]]
exec(compile(src, "profile_post2.py", "exec"))
PY
rmdir profiles_tmp 2>/dev/null
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/mfma $OUT/stats
ls -la gpurun_out/profiles_r4
