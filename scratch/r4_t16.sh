#!/bin/bash
# round 4, trip 16: profile sets of the final library (ABI 11, 128 x 128 kernel with the batched epilogue)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash scratch/profile_round4.sh r4i_cls r4_i_cls_bs64 > gpurun_out/r4_prof_clsi.log 2>&1 || { tail -20 gpurun_out/r4_prof_clsi.log; exit 1; }
echo cls done
bash scratch/profile_round4.sh r4i_mae r4_i_mae_bs256 --workload mae > gpurun_out/r4_prof_maei.log 2>&1 || { tail -20 gpurun_out/r4_prof_maei.log; exit 1; }
echo mae done
bash scratch/profile_round4.sh r4i_mae64 r4_i_mae_bs64 --workload mae --batch 64 > gpurun_out/r4_prof_mae64i.log 2>&1 || { tail -20 gpurun_out/r4_prof_mae64i.log; exit 1; }
ls gpurun_out/profiles_r4 | grep r4_i
