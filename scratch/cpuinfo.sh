#!/bin/bash
echo "nproc: $(nproc)"; lscpu | head -20; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os,torch; print('affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads())"; free -g | head -2
