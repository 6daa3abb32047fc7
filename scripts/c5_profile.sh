#!/bin/bash
# dev tool (GPU box): one -medoidtree -gt upgma run at N family sequences with the stage timers (-v) and the
# thread-second accounts of the recursion (FAMSA_GPU_PROFILE=1) -> gpurun_out/c5_profile.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
N=${1:-3000000}
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
f = "/tmp/fam_$N.fasta"
if not os.path.exists(f):
    seqio.family_fasta($N, 300, f)
PY
TIMEFORMAT='shell: wall=%R s user=%U s sys=%S s'
for rep in 1 2; do
  { time FAMSA_GPU_PROFILE=1 timeout 300 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/fam_$N.fasta /tmp/fam_$N.dnd 2> gpurun_out/c5_profile.txt ; } 2> /tmp/c5_shell_time.txt
  cat /tmp/c5_shell_time.txt >> gpurun_out/c5_profile.txt
done
# the same without the recursion's accounts (FAMSA_GPU_PROFILE keeps the engine alive until its report is printed)
{ time timeout 300 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/fam_$N.fasta /tmp/fam_$N.dnd 2> /tmp/c5_plain.txt ; } 2> /tmp/c5_shell_time.txt
{ echo "--- without FAMSA_GPU_PROFILE"; cat /tmp/c5_plain.txt /tmp/c5_shell_time.txt; } >> gpurun_out/c5_profile.txt
cat gpurun_out/c5_profile.txt
