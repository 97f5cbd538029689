#!/bin/bash
# round-3 experiment 19: stream -> hardware-queue aliasing.  The default bench builds several models in one process; with a stream set
# per model the MAE step of that line lost 11 % (28.5 vs 25.6 ms).  Shared side streams, and GPU_MAX_HW_QUEUES 4 (default) vs 8.
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-c5 --no-kernel-stats"
sel='import json,sys; r=json.loads(sys.stdin.read()); print("cls", r["value"], "mae", r["mae"]["value"], {k: v.get("value") for k, v in r.get("finetune_modes", {}).items() if isinstance(v, dict)})'
for q in "" 8 "" 8; do echo -n "GPU_MAX_HW_QUEUES=${q:-default}: "; if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi; python bench.py $F 2>/dev/null | python -c "$sel"; done
