#!/bin/bash
# usage: pmc_gemm.sh <which> <cfg>   (two counter passes, per-dispatch CSV -> gpurun_out/pmc_<which>_<cfg>_*.csv)
cd /tmp && export TMPDIR=/tmp
W=$1; C=$2; OUT=/root/repo/gpurun_out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmcA -o a_${W}_${C} --output-format csv -- python3 /root/repo/scratch/one_gemm.py $W $C > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/pmcB -o b_${W}_${C} --output-format csv -- python3 /root/repo/scratch/one_gemm.py $W $C > $OUT/pmcB.log 2>&1
ls $OUT/pmcA $OUT/pmcB
