#!/bin/bash
# round 4, trip 2: first run of precision mode fp16 on the GPU (op tests over the three modes, fp16 end-to-end tests, fp16 parity at ViT-B)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp16.py -q -m gpu -x -s > gpurun_out/r4_t2_ops.log 2>&1
RC=$?
tail -5 gpurun_out/r4_t2_ops.log
[ $RC -eq 0 ] || { grep -E "Error|assert|FAILED" gpurun_out/r4_t2_ops.log | head -30; exit $RC; }
timeout -k 10 800 python -m pytest tests/test_gpu_parity_large.py -q -m gpu -s -k "fp16" > gpurun_out/r4_t2_parity.log 2>&1
RC=$?
grep -E "\[parity\]|\[measured\]|passed|failed|Error" gpurun_out/r4_t2_parity.log | tail -40
exit $RC
