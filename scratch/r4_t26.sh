#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_models.py -q -m gpu -x -s -k "top_block" 2>&1 | grep -E "measured\] cls-row|passed|failed|rror|assert" | tail -16
