#!/bin/bash
# dev tool (GPU box): where the MedoidTree tree stage spends its CLARANS time -> gpurun_out/clarans_profile.txt
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
OUT=gpurun_out/clarans_profile.txt
FAMSA_GPU_CLEAN_EXIT=1 LCSGPU_PROFILE=1 FAMSA_GPU_PROFILE=1 timeout 300 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/p.dnd 2> $OUT
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cprof
FAMSA_GPU_CLEAN_EXIT=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/cprof -o run -- $ROOT/famsa_amd/famsa-gpu -medoidtree -gt upgma -gt_export $F /tmp/p2.dnd > /tmp/cprof.log 2>&1
tail -3 /tmp/cprof.log
python $ROOT/scripts/rocpd_summary.py $(find /tmp/cprof -name "*.db") > $ROOT/gpurun_out/clarans_kernel_stats.txt
python $ROOT/scripts/rocpd_timeline.py $(find /tmp/cprof -name "*.db") clarans >> $ROOT/gpurun_out/clarans_kernel_stats.txt 2>&1
python $ROOT/scripts/rocpd_timeline.py $(find /tmp/cprof -name "*.db") lcs_rows >> $ROOT/gpurun_out/clarans_kernel_stats.txt 2>&1
python $ROOT/scripts/rocpd_timeline.py $(find /tmp/cprof -name "*.db") lcsgpu >> $ROOT/gpurun_out/clarans_kernel_stats.txt 2>&1
