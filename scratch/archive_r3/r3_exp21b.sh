#!/bin/bash
# kernel + memory-copy trace of the host-input step with 4 and 8 hardware queues (which streams share a queue?)
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 6 --warmup 3 --preheat 0.3 --input host"
REPO=$PWD; cd /tmp && export TMPDIR=/tmp
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  OUT=$REPO/gpurun_out/prof_hi$q; mkdir -p $OUT
  rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/trace -o t -- python3 $REPO/bench.py $F > $OUT/trace.log 2>&1
  python3 $REPO/scratch/trace_timeline.py $OUT/trace/t_results.db -2 0,40 > $REPO/gpurun_out/r3_exp21_timeline_q$q.txt 2>&1
  python3 - $OUT/trace/t_results.db > $REPO/gpurun_out/r3_exp21_copies_q$q.txt 2>&1 <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
mc = [t for t in tabs if 'memory_copy' in t]
print(mc)
for t in mc:
    cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
    print(t, cols)
    rows = list(c.execute(f"select * from {t} order by start desc limit 12"))
    for r in rows: print(r)
PY
  rm -rf $OUT/trace
done
