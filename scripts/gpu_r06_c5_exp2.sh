#!/bin/bash
# round 6: the FastTree recursion level by level at 3 x 10^6 sequences -- CU partition on / off, timings per level
cd "$(dirname "$0")/.."
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
if not os.path.exists("/tmp/f3m.fasta"):
    seqio.family_fasta(3000000, 300, "/tmp/f3m.fasta")
PY
for mode in ${MODES:-mask nomask}; do
for rep in 1 2; do
echo "== $mode"
if [ $mode = nomask ]; then export LCSGPU_NO_CU_MASK=1; else unset LCSGPU_NO_CU_MASK; fi
LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/f3m.fasta /tmp/o.dnd 2>&1 | grep -E "fasttree.level [0-9]|fasttree.tail|tree_build|triangles_batch:|clarans.batch chains|clarans.batch parts"
sha256sum /tmp/o.dnd | cut -c1-16
done
done
