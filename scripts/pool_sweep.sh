#!/bin/bash
# dev tool (GPU box): MedoidTree tree stage of the 3 x 10^6 family set against the number of pool threads / lanes
# usage: scripts/pool_sweep.sh [n_seqs]   -> gpurun_out/pool_sweep.txt
N=${1:-3000000}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/pool_sweep.txt
: > $OUT
F=/tmp/family_${N}_300.fasta
python - <<PY
import sys, os, time
sys.path.insert(0, '.')
from famsa_amd import seqio
t0 = time.time()
if not os.path.exists("$F"):
    seqio.family_fasta($N, 300, "$F")
print("fasta %.1f s" % (time.time() - t0))
PY
run() { # label, env...
  label=$1; shift
  for rep in 1 2; do
    t0=$(date +%s%N)
    env "$@" FAMSA_GPU_PROFILE=1 timeout 300 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sweep.dnd 2> /tmp/sweep.err
    rc=$?
    t1=$(date +%s%N)
    echo "$label rep=$rep rc=$rc wall_ms=$(( (t1 - t0) / 1000000 )) sha=$(sha256sum /tmp/sweep.dnd | cut -c1-16) $(grep -E 'tree|clarans|partial|lcs_calls|assign' /tmp/sweep.err | tr '\n' ' ')" >> $OUT
  done
}
run "defaults (priority streams)" X=1
run "no priority" LCSGPU_CLARANS_NO_PRIORITY=1
run "defaults (priority streams) again" X=1
cat $OUT
