#!/bin/bash
# round-3 experiment 4: dispatcher rules (dGELU -> cfg 8; tall M -> staged epilogues) in-step, and the half-batch MAE decoder shapes
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"], {k:(v["avg_us"]) for k,v in r["roofline"]["hbm_kernels"].items()}, {k:v["avg_us"] for k,v in r["roofline"]["gemm_by_layout"].items()})'
echo "== cfg sweep MAE decoder half batch M=25216 D=512 =="; M=25216 D=512 CFGS=0,8,9,10,24,25,26 python scratch/bench_gemm6.py 2>&1 | grep -v amdgpu | tail -8
for i in 1 2; do
echo "== cls =="; python bench.py $F 2>/dev/null | python -c "$sel"
echo "== MAE tall=32768 =="; python bench.py --workload mae $F 2>/dev/null | python -c "$sel"
echo "== MAE tall=20000 =="; PM_TALL_M=20000 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"
echo "== MAE tall off =="; PM_TALL_M=100000000 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"
done
