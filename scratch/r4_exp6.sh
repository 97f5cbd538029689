#!/bin/bash
# round 4, experiment 6: 128 x 256 ring tiles for the half-batch f32-residual forward GEMMs (proj / fc2 at M = 6 304: 99 tiles of 192 rows today)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_exp6_standalone.txt
import os, sys, torch
sys.path.insert(0, os.getcwd())
from ssl4polyp_amd.engine import Kernels
from ssl4polyp_amd._lib import EPI_RESIDUAL
dev = "cuda"; bf = torch.bfloat16
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
def run(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
D = 768
for M in (6304, 6400, 12608):
    for name, N, K in (("proj", D, D), ("fc2", D, 4 * D)):
        x, W, bias = t(M, K), t(N, K), torch.zeros(N, device=dev)
        out, res = torch.empty(M, N, device=dev), t(M, N, dt=torch.float32)
        ref = x.float() @ W.float().t() + res
        row = []
        for v in (0, 12, 13, 1):
            k = Kernels("bf16"); k.gemm_variant = v
            fn = lambda: k.linear_fwd(x, W, bias, out, M, N, K, EPI_RESIDUAL, resid=res)
            us = run(fn)
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            row.append(f"cfg{v} {us:.1f}us (err {err:.1e})")
        print(f"M={M} {name}: " + "  ".join(row))
PY
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-c5 --no-mae"
for rep in 1 2; do
for cfg in 0 12 13; do
  for wl in "cls 64" "mae 256"; do
    set -- $wl
    PM_CFG_CLASS="0,0,$cfg,0,0" timeout -k 10 200 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_exp6_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp6_tmp.json')); print('nt_residual cfg $cfg rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done; done | tee gpurun_out/r4_exp6_residual_tiles.txt
