#!/bin/bash
# dev tool (GPU box): what the end of a 3 000 000-record run costs: wall clock of the shell against the tool's own clock up to
# its exit, with the fast exit (default) and with the orderly one (FAMSA_GPU_CLEAN_EXIT=1) -> gpurun_out/exit_cost.txt
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
OUT=gpurun_out/exit_cost.txt
: > $OUT
TIMEFORMAT='wall=%R user=%U sys=%S'
for mode in fast clean; do
  if [ $mode = clean ]; then export FAMSA_GPU_CLEAN_EXIT=1; else unset FAMSA_GPU_CLEAN_EXIT; fi
  sleep 3
  { time timeout 300 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/fam_$N.fasta /tmp/fam_$N.dnd 2> /tmp/ec.err ; } 2> /tmp/ec.time
  echo "$mode: $(cat /tmp/ec.time) $(grep -E 'main_until_exit|tree_build|mem\.' /tmp/ec.err | tr '\n' ' ')" >> $OUT
done
unset FAMSA_GPU_CLEAN_EXIT
cat $OUT
