#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmcA -o a_attn --output-format csv -- python3 /root/repo/scratch/one_attn.py > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace -d $OUT/pmcB -o b_attn --output-format csv -- python3 /root/repo/scratch/one_attn.py > $OUT/pmcB.log 2>&1
