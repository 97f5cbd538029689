#!/bin/bash
# round 4, trip 18: smoke() and the CPU-side checks on the GPU box with the final library
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4
timeout -k 10 600 python -m pytest tests -x -q -m "not gpu" 2>&1 | tail -2
