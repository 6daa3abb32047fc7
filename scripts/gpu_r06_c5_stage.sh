#!/bin/bash
# round 6 (second session): where the tree stage of -medoidtree -gt upgma at 3 x 10^6 sequences goes outside the levels
# (fasttree.stage, newick lines under LCSGPU_PROFILE), then plain runs with the residues released early / late / never
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
R=gpurun_out/c5_stage_r06.txt
: > $R
for n in 3000000 1000000; do
WANT=$(python -c "import json; print(json.load(open('tests/golden/meta_large.json')).get('family$n', {}).get('medoid_upgma_newick_sha256', 'no-pin'))")
for rep in 1 2; do
  echo "== $n profiled run $rep" >> $R
  LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/family_${n}_300.fasta /tmp/o.dnd 2>&1 | grep -E "fasttree.stage|fasttree.level [0-9]|fasttree.level parts|fasttree.tail|newick:|tree stage:|^time\." >> $R
done
for how in late no_level_scratch late no_level_scratch late no_level_scratch; do
for rep in 1 2 3; do
  t0=$(date +%s.%N)
  FAMSA_HOST_TEST=$how famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/family_${n}_300.fasta /tmp/o.dnd 2> /tmp/o.err
  t1=$(date +%s.%N)
  echo "family$n $how $(grep -E 'time.tree_build|time.newick|time.main_until_exit' /tmp/o.err | tr '\n' ' ') wall=$(python -c "print(round($t1-$t0,3))") newick=$([ "$(sha256sum /tmp/o.dnd | cut -d' ' -f1)" = "$WANT" ] && echo identical-to-the-reference || echo DIFFERENT)" >> $R
done
done
done
cat $R
