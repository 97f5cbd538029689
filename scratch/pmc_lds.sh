#!/bin/bash
# usage (GPU box, repo root): bash scratch/pmc_lds.sh <tag> [bench args...]
# LDS-side counters per kernel of the benched step (own pass, counters + --kernel-trace only)
TAG=$1; shift
REPO=$PWD; OUT=$REPO/gpurun_out/pmc_lds_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT -o l --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 2 --preheat 0 --no-cpu-baseline --no-kernel-stats --no-parity --no-torch-baseline --no-mae "$@" > $OUT/run.log 2>&1
cd $REPO && python3 - $OUT/l_counter_collection.csv > $OUT/summary.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))
print("# kernel | launches | LDS bank-conflict cycles / LDS active cycles | LDS active / CU-busy cycles | data-FIFO-full / active | cmd-FIFO-full / active | wait-LDS / CU-busy")
for k, c in rows[:16]:
    act = max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1); busy = max(c.get("SQ_BUSY_CU_CYCLES", 0), 1)
    print(f"{k:70s} | {n[k]:4d} | {c.get('SQ_LDS_BANK_CONFLICT',0)/act:6.3f} | {act/busy:6.3f} | {c.get('SQ_LDS_DATA_FIFO_FULL',0)/act:6.3f} | {c.get('SQ_LDS_CMD_FIFO_FULL',0)/act:6.3f} | {c.get('SQ_WAIT_INST_LDS',0)/busy:6.3f}")
PY
cat $OUT/summary.txt
