#!/bin/bash
# round 4, trip 5: the final default bench line with clock / power / memory-busy telemetry beside it, then the MAE bs=64 profile set again (few-tiles rule)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python scratch/telemetry.py gpurun_out/r4_t5_telemetry.csv --hz 50 --seconds 500 &
TPID=$!
sleep 1
timeout -k 10 500 python bench.py --steps 40 --warmup 10 > gpurun_out/r4_t5_bench.json 2> gpurun_out/r4_t5_bench.err
RC=$?
kill $TPID; wait $TPID
echo "bench rc $RC"; tail -3 gpurun_out/r4_t5_bench.err
[ $RC -eq 0 ] || exit $RC
bash scratch/profile_round4.sh r4g_mae64 r4_g_mae_bs64 --workload mae --batch 64 > gpurun_out/r4_prof_mae64g.log 2>&1 || { tail -20 gpurun_out/r4_prof_mae64g.log; exit 1; }
ls gpurun_out/profiles_r4 | grep r4_g
