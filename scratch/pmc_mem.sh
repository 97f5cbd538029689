#!/bin/bash
cd /tmp && export TMPDIR=/tmp
W=$1; C=$2; OUT=/root/repo/gpurun_out
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace -d $OUT/pmcC -o c_${W}_${C} --output-format csv -- python3 /root/repo/scratch/one_gemm.py $W $C > $OUT/pmcC.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum --kernel-trace -d $OUT/pmcD -o d_${W}_${C} --output-format csv -- python3 /root/repo/scratch/one_gemm.py $W $C > $OUT/pmcD.log 2>&1
tail -3 $OUT/pmcC.log
