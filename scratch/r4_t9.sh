#!/bin/bash
# round 4, trip 9: final state (few-tiles rule + narrow N, dh=32 pair map, banded tile order): whole GPU suite, default bench line, profile sets
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4_t9_pytest.log 2>&1
RC=$?
tail -3 gpurun_out/r4_t9_pytest.log
[ $RC -eq 0 ] || { grep -E "Error|assert|FAILED" gpurun_out/r4_t9_pytest.log | head -20; exit $RC; }
timeout -k 10 500 python bench.py > gpurun_out/r4_t9_bench.json 2> gpurun_out/r4_t9_bench.err
RC=$?
echo "bench rc $RC"; tail -3 gpurun_out/r4_t9_bench.err
[ $RC -eq 0 ] || exit $RC
bash scratch/profile_round4.sh r4h_cls r4_h_cls_bs64 > gpurun_out/r4_prof_clsh.log 2>&1 || { tail -20 gpurun_out/r4_prof_clsh.log; exit 1; }
bash scratch/profile_round4.sh r4h_mae r4_h_mae_bs256 --workload mae > gpurun_out/r4_prof_maeh.log 2>&1 || { tail -20 gpurun_out/r4_prof_maeh.log; exit 1; }
bash scratch/profile_round4.sh r4h_mae64 r4_h_mae_bs64 --workload mae --batch 64 > gpurun_out/r4_prof_mae64h.log 2>&1 || { tail -20 gpurun_out/r4_prof_mae64h.log; exit 1; }
ls gpurun_out/profiles_r4 | grep r4_h
