#!/bin/bash
# round 4, trip 22: sparse top block -- equality test against the dense backward, then the suites that exercise the classifier
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_models.py -q -m gpu -x -s -k "top_block" 2>&1 | grep -E "measured\] sparse|passed|failed|rror" | tail -14
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4_t22_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r4_t22_pytest.log
[ $rc -eq 0 ] || { grep -E "Error|assert|FAILED" gpurun_out/r4_t22_pytest.log | head -20; exit $rc; }
