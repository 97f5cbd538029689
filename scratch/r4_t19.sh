#!/bin/bash
# round 4, trip 19: the eval-time perturbations on the device (DevicePerturber): bit-exactness against the reference's outputs, rate
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_augment.py tests/test_gpu_input.py -q -m gpu -x > gpurun_out/r4_t19_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r4_t19_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scratch/bench_perturb.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_t19_perturb_rate.txt
