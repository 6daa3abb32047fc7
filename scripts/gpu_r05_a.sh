#!/bin/bash
# round 5, GPU call A: the CLARANS tests first (the pruned kernels + the filtered evaluation), C5 at 3 000 000 sequences
# timed and traced, then the whole GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_clarans.py -x -q -m gpu -k "matches_reference or concurrent or refused or rejects" > gpurun_out/a_clarans_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/a_clarans_tests.txt
tail -15 gpurun_out/a_clarans_tests.txt
bash scripts/c5_profile.sh 3000000 > /dev/null 2>&1
bash scripts/clarans_profile.sh 3000000 > /dev/null 2>&1
head -40 gpurun_out/clarans_kernel_stats.txt
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/a_suite.txt 2>&1
tail -15 gpurun_out/a_suite.txt
grep -E "tree_build|sha256|clarans.searches|shell" gpurun_out/c5_profile.txt
