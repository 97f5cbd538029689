#!/bin/bash
# round 4, trip 1: clock / power telemetry beside the default bench line, then the in-kernel clock inside the step (stamp build)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python scratch/telemetry.py gpurun_out/r4_t1_telemetry.csv --hz 50 --seconds 500 &
TPID=$!
sleep 1
timeout -k 10 420 python bench.py --steps 100 --warmup 20 > gpurun_out/r4_t1_bench.json 2> gpurun_out/r4_t1_bench.err
RC=$?
kill $TPID; wait $TPID
echo "bench rc $RC"
[ $RC -eq 0 ] || exit $RC
POLYPMAE_LIB=$PWD/ssl4polyp_amd/lib/libpolypmae_stamp.so timeout -k 10 240 python scratch/inkernel_clock.py cls mae qkv zeros > gpurun_out/r4_t1_inkernel_clock.json 2> gpurun_out/r4_t1_inkernel_clock.err
echo "clock rc $?"
tail -c 1500 gpurun_out/r4_t1_inkernel_clock.json
