#!/bin/bash
# GPU box: kernel trace (rocprofv3 --kernel-trace --stats) of the -medoidtree -gt upgma run at N family sequences: per-kernel
# totals, the CLARANS rounds and the LCS launches by duration / grid / queue, how many are in flight together
# -> gpurun_out/clarans_kernel_stats.txt
N=${1:-3000000}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F=/tmp/family_${N}_300.fasta
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
if not os.path.exists("$F"):
    seqio.family_fasta($N, 300, "$F")
PY
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cprof
FAMSA_GPU_CLEAN_EXIT=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/cprof -o run -- $ROOT/famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/p2.dnd > /tmp/cprof.log 2>&1
grep "time.tree_build\|gpu.lcs_kernel_ms" /tmp/cprof.log > $ROOT/gpurun_out/clarans_kernel_stats.txt
python $ROOT/scripts/rocpd_summary.py $(find /tmp/cprof -name "*.db") >> $ROOT/gpurun_out/clarans_kernel_stats.txt
python $ROOT/scripts/rocpd_timeline.py $(find /tmp/cprof -name "*.db") clarans >> $ROOT/gpurun_out/clarans_kernel_stats.txt 2>&1
python $ROOT/scripts/rocpd_timeline.py $(find /tmp/cprof -name "*.db") lcs_rows >> $ROOT/gpurun_out/clarans_kernel_stats.txt 2>&1
