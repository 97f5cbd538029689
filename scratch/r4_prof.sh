#!/bin/bash
# round 4: rocprofv3 summaries of the final default path (kernel stats, timeline, FETCH / WRITE PMC, MFMA-busy PMC) per workload
cd "$GRAFT_REPO_ROOT" || exit 1
bash scratch/profile_round4.sh r4f_cls r4_f_cls_bs64 > gpurun_out/r4_prof_cls.log 2>&1 || { tail -20 gpurun_out/r4_prof_cls.log; exit 1; }
echo cls done
bash scratch/profile_round4.sh r4f_mae r4_f_mae_bs256 --workload mae > gpurun_out/r4_prof_mae.log 2>&1 || { tail -20 gpurun_out/r4_prof_mae.log; exit 1; }
echo mae done
bash scratch/profile_round4.sh r4f_cls16 r4_f_cls_bs64_fp16 --precision fp16 > gpurun_out/r4_prof_cls16.log 2>&1 || { tail -20 gpurun_out/r4_prof_cls16.log; exit 1; }
echo cls fp16 done
bash scratch/profile_round4.sh r4f_mae64 r4_f_mae_bs64 --workload mae --batch 64 > gpurun_out/r4_prof_mae64.log 2>&1 || { tail -20 gpurun_out/r4_prof_mae64.log; exit 1; }
echo mae64 done
ls gpurun_out/profiles_r4
