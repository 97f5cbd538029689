#!/bin/bash
# usage (GPU box, repo root): bash scratch/pmc_mfma.sh <tag> [bench args...]
# MFMA-pipe utilisation per kernel of the benched step: rocprofv3 --pmc (counters only, with --kernel-trace) over bench.py.
TAG=$1; shift
REPO=$PWD; OUT=$REPO/gpurun_out/pmc_mfma_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT -o m --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-stats --no-parity --no-torch-baseline --no-mae "$@" > $OUT/run.log 2>&1
cd $REPO && python3 scratch/pmc_mfma_post.py $OUT/m_counter_collection.csv > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
