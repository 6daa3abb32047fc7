#!/bin/bash
# round 6 (second session): the levels' own LCS launches alone on the chip (LCSGPU_TUNE front_alone=1) against beside the leaf batches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - <<PY
import sys, os
sys.path.insert(0, ".")
from famsa_amd import seqio
for n in (1000000, 3000000):
    p = "/tmp/family_%d_300.fasta" % n
    if not os.path.exists(p):
        seqio.family_fasta(n, 300, p)
PY
R=gpurun_out/front_r06.txt
: > $R
for rep in 1 2 3; do
for how in front_alone=0 front_alone=1; do
for n in 3000000 1000000; do
for k in 1 2 3; do
  WANT=$(python -c "import json; print(json.load(open('tests/golden/meta_large.json')).get('family$n', {}).get('medoid_upgma_newick_sha256', 'no-pin'))")
  LCSGPU_TUNE=$how famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/family_${n}_300.fasta /tmp/o.dnd 2> /tmp/o.err
  echo "family$n $how $(grep -E 'time.tree_build|time.main_until_exit' /tmp/o.err | tr '\n' ' ') newick=$([ "$(sha256sum /tmp/o.dnd | cut -d' ' -f1)" = "$WANT" ] && echo identical || echo DIFFERENT)" >> $R
done
done
done
done
for how in front_alone=0 front_alone=1; do
  echo "== $how, profiled" >> $R
  LCSGPU_TUNE=$how LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/family_3000000_300.fasta /tmp/o.dnd 2>&1 | grep -E "fasttree.stage|fasttree.level [0-9]|fasttree.tail|clarans.batch parts|level parts|tree_build" >> $R
done
