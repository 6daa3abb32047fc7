#!/bin/bash
# dev tool (GPU box): 3 x 10^6-sequence MedoidTree tree stage against pool threads x CLARANS batch streams (round 4's
# one-launch rounds) -> gpurun_out/pool_sweep2.txt
N=${1:-3000000}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/pool_sweep2.txt
: > $OUT
F=/tmp/fam_$N.fasta
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
if not os.path.exists("$F"):
    seqio.family_fasta($N, 300, "$F")
PY
for rep in 1 2; do
  for cfg in "32 4" "48 4" "64 4" "32 8" "64 8" "96 8"; do
    set -- $cfg
    FAMSA_GPU_POOL_THREADS=$1 LCSGPU_CLARANS_GROUPS=$2 FAMSA_GPU_PROFILE=1 timeout 120 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sweep.dnd 2> /tmp/sweep.err
    echo "pool=$1 groups=$2 rep=$rep rc=$? $(grep -E 'time.tree_build|engine.clarans' /tmp/sweep.err | tr '\n' ' ') sha=$(sha256sum /tmp/sweep.dnd | cut -c1-12)" >> $OUT
  done
done
cat $OUT
