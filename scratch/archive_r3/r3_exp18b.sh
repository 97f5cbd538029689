#!/bin/bash
# is the two-launch weight-gradient schedule (exp 18) deterministic?  per-step gradient hashes, MAE B=256 and cls B=64
for v in 0 1; do for wl in mae cls; do echo "== PM_TWO_GROUPS=$v $wl"; PM_TWO_GROUPS=$v python scratch/det_mae.py $wl 16 2>&1 | grep -v "amdgpu.ids\|Warning\|detach"; done; done
