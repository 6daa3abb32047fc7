#!/bin/bash
# dev tool (GPU box): the UPGMA merge phase -- batches of 32 / 16 / 8 merges per launch pair (upgma_batch_kernels.hip)
# against one launch per merge (LCSGPU_UPGMA_BATCH=0), uniform and family sets; prints lcsgpu_upgma's own timing lines.
python - <<PY
import sys; sys.path.insert(0,".")
from famsa_amd import seqio
for n in (10000, 100000):
    c,o=seqio.synth_uniform(n,400); seqio.to_fasta(c,o,"/tmp/u_%d.fasta" % n)
seqio.family_fasta(30000, 300, "/tmp/f_30000.fasta")
PY
for f in /tmp/u_10000.fasta /tmp/f_30000.fasta /tmp/u_100000.fasta; do
for cfg in "LCSGPU_UPGMA_BATCH=32" "LCSGPU_UPGMA_BATCH=16" "LCSGPU_UPGMA_BATCH=8" "LCSGPU_UPGMA_BATCH=0"; do
  for gt in upgma upgma_modified; do
   echo "$f $gt $cfg:"
   env $cfg LCSGPU_PROFILE=1 timeout 300 famsa_amd/famsa-gpu -gt $gt -gt_export $f /tmp/u_$gt.dnd 2>&1 | grep lcsgpu_upgma | sed 's/^/    /'
   sha256sum /tmp/u_$gt.dnd | cut -c1-16 | sed 's/^/    newick /'
  done
done
done
