#!/bin/bash
# stand-alone kernel sweep after the vmcnt-guard fix: new library vs the guarded build (libpolypmae_alt.so, -DPM_AUTO_VMCNT)
O=gpurun_out/sweep_r2d; mkdir -p $O; L=$PWD/ssl4polyp_amd/lib/libpolypmae_alt.so
python scratch/bench_attn.py > $O/attn_new.txt 2>&1
PM_ATTN_FWD2_DH32=1 python scratch/bench_attn.py > $O/attn_new_fwd2dh32.txt 2>&1
POLYPMAE_LIB=$L python scratch/bench_attn.py > $O/attn_guarded.txt 2>&1
CFGS=0,6,9,10,25,26 python scratch/bench_gemm6.py > $O/gemm768_new.txt 2>&1
POLYPMAE_LIB=$L CFGS=0,6,9,10,25,26 python scratch/bench_gemm6.py > $O/gemm768_guarded.txt 2>&1
M=50432 D=512 CFGS=0,6,9,10,25,26 python scratch/bench_gemm6.py > $O/gemm512_new.txt 2>&1
for wv in 0 2 3; do WV=$wv python scratch/bench_wgrad.py > $O/wgrad_new_wv$wv.txt 2>&1; done
POLYPMAE_LIB=$L python scratch/bench_wgrad.py > $O/wgrad_guarded.txt 2>&1
tail -n 20 $O/*.txt
