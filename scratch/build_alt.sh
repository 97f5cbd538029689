#!/bin/bash
# usage: bash scratch/build_alt.sh "<extra hipcc flags>"   -> ssl4polyp_amd/lib/libpolypmae_alt.so (A/B builds: POLYPMAE_LIB=...)
python3 - "$@" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
g.build_library(force=True, lib=os.path.join(os.path.dirname(g.LIB), "libpolypmae_alt.so"), extra_flags=sys.argv[1].split())
print("built alt library with", sys.argv[1])
PY
