#!/bin/bash
# GPU box: -medoidtree -gt upgma at N family sequences (default 3 000 000 = BASELINE config C5) with the stage timers (-v),
# the thread-second accounts of the recursion and the engine's search statistics (LCSGPU_PROFILE=1), twice, then once plain;
# every Newick's sha256 next to the reference's (tests/golden/meta_large.json) -> gpurun_out/c5_profile.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
N=${1:-3000000}
F=/tmp/family_${N}_300.fasta
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
if not os.path.exists("$F"):
    seqio.family_fasta($N, 300, "$F")
PY
WANT=$(python -c "import json; print(json.load(open('tests/golden/meta_large.json')).get('family$N', {}).get('medoid_upgma_newick_sha256', 'no-pin'))")
OUT=gpurun_out/c5_profile.txt
: > $OUT
TIMEFORMAT='shell: wall=%R s user=%U s sys=%S s'
for rep in 1 2; do
  { time LCSGPU_PROFILE=1 timeout 300 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/fam_$N.dnd 2> /tmp/c5_err.txt ; } 2> /tmp/c5_shell_time.txt
  { echo "--- run $rep (LCSGPU_PROFILE=1)"; grep -v "^lcsgpu_create" /tmp/c5_err.txt; cat /tmp/c5_shell_time.txt; echo "newick sha256 $(sha256sum /tmp/fam_$N.dnd | cut -d' ' -f1) reference $WANT"; } >> $OUT
done
{ time timeout 300 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/fam_$N.dnd 2> /tmp/c5_plain.txt ; } 2> /tmp/c5_shell_time.txt
{ echo "--- plain"; cat /tmp/c5_plain.txt /tmp/c5_shell_time.txt; echo "newick sha256 $(sha256sum /tmp/fam_$N.dnd | cut -d' ' -f1) reference $WANT"; } >> $OUT
cat $OUT
