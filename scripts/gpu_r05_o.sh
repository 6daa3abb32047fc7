#!/bin/bash
# round 5, GPU call O: the final code once more -- the GPU suite, the kernel trace of C5 (rocprofv3 --kernel-trace)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=5 ) > gpurun_out/o_suite.txt 2>&1
tail -16 gpurun_out/o_suite.txt
bash scripts/clarans_profile.sh > /dev/null 2>&1; cp gpurun_out/clarans_kernel_stats.txt gpurun_out/o_c5_kernels.txt
head -24 gpurun_out/o_c5_kernels.txt | cut -c1-180
