#!/bin/bash
# round-3 experiment 7: one-block-per-head attention forward for the half-batch chains (PM_ATTN_V1), host under load
python -m pytest tests/test_gpu_augment.py -m gpu -x -q 2>&1 | tail -3
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"], {k:v["avg_us"] for k,v in r["roofline"]["attention"].items()}, r.get("busy_host"))'
for i in 1 2; do
echo "== cls default =="; python bench.py $F 2>/dev/null | python -c "$sel"
echo "== cls PM_ATTN_V1=1 =="; PM_ATTN_V1=1 python bench.py $F 2>/dev/null | python -c "$sel"
done
echo "== MAE default =="; python bench.py --workload mae $F 2>/dev/null | python -c "$sel"
echo "== MAE PM_ATTN_V1=1 =="; PM_ATTN_V1=1 python bench.py --workload mae $F 2>/dev/null | python -c "$sel"
echo "== cls host busy 8 =="; python bench.py $F --host-busy 8 2>/dev/null | python -c "$sel"
echo "== cls host busy 16 =="; python bench.py $F --host-busy 16 2>/dev/null | python -c "$sel"
echo "== MAE host busy 8 =="; python bench.py --workload mae $F --host-busy 8 2>/dev/null | python -c "$sel"
python scratch/bench_input_aug.py 64 2>&1 | grep -v "amdgpu\|Warning\|torch.from_numpy"
