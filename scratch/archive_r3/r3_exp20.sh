#!/bin/bash
# round-3 experiment 20: the data-parallel stream schedule on ONE GPU (bench.py --force-sync: bucketed RCCL all-reduces at world size 1
# on the comm stream + RCCL's own streams) against the plain step -- engine streams reserved before RCCL initialises (default) or
# created at first use (BENCH_LATE_STREAMS=1: two of them share a hardware queue), 4 (default) and 8 hardware queues; then a kernel
# trace of the forced-sync step (per-queue busy time and gaps)
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"], r["host_enqueue_ms_per_step"], r["config"]["parallelism"])'
run() { echo -n "$2 queues=${1:-default} $3: "; if [ -n "$1" ]; then export GPU_MAX_HW_QUEUES=$1; else unset GPU_MAX_HW_QUEUES; fi; python bench.py $F --workload $2 $3 2>>gpurun_out/r3_exp20.err | python -c "$sel"; }
for wl in cls mae; do
run "" $wl ""
run "" $wl --force-sync          # (bench.py itself asks for 8 queues when it runs data-parallel)
run 4 $wl --force-sync
BENCH_LATE_STREAMS=1 run "" $wl --force-sync
run 16 $wl --force-sync
run "" $wl --force-sync
done
unset GPU_MAX_HW_QUEUES
REPO=$PWD; OUT=$REPO/gpurun_out/prof_fs; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o t -- python3 $REPO/bench.py --steps 6 --warmup 3 $F --preheat 0.3 --force-sync > $OUT/trace.log 2>&1
cd $REPO
python3 scratch/trace_timeline.py $OUT/trace/t_results.db -2 > gpurun_out/r3_exp20_timeline.txt 2>&1
python3 scratch/trace_timeline.py $OUT/trace/t_results.db -2 0,40 > gpurun_out/r3_exp20_dump.txt 2>&1
rm -rf $OUT/trace
