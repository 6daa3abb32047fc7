#!/bin/bash
# round 5, GPU call G: the whole suite with the one-workgroup searches and the leaf-first pool; C5 profile (stage timers,
# timeline, searches), the CLARANS / LCS kernels of that run by rocprofv3, five plain C5 runs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=25 ) > gpurun_out/g_suite.txt 2>&1
tail -40 gpurun_out/g_suite.txt
bash scripts/c5_profile.sh > /dev/null 2>&1; cp gpurun_out/c5_profile.txt gpurun_out/g_c5_profile.txt
grep -E "tree_build|sha256|fasttree.top|engine.clarans" gpurun_out/g_c5_profile.txt
F=/tmp/family_3000000_300.fasta
: > gpurun_out/g_c5_runs.txt
for rep in 1 2 3 4 5; do
  famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> /tmp/sw.err
  echo "run $rep $(grep -E 'time.tree_build|gpu.lcs_kernel_ms|time.main_until_exit' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/g_c5_runs.txt
done
cat gpurun_out/g_c5_runs.txt
bash scripts/clarans_profile.sh > /dev/null 2>&1; cp gpurun_out/clarans_kernel_stats.txt gpurun_out/g_c5_kernels.txt
head -40 gpurun_out/g_c5_kernels.txt | cut -c1-180
