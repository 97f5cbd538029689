#!/bin/bash
# usage (GPU box, repo root): bash scratch/pmc_attn.sh <tag> [ENV=...]   -- two PMC passes over scratch/one_attn.py
TAG=$1; shift
for e in "$@"; do export "$e"; done
R=$PWD; OUT=$R/gpurun_out/pmc_attn_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/a -o a --output-format csv -- python3 $R/scratch/one_attn.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace -d $OUT/b -o b --output-format csv -- python3 $R/scratch/one_attn.py > $OUT/b.log 2>&1
cd $R; python3 scratch/pmc_read.py $OUT/a/a_counter_collection.csv $OUT/b/b_counter_collection.csv > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
