#!/bin/bash
# dev tool (GPU box): UPGMA merge phase, one launch per merge vs all merges in one kernel on one XCD, both layouts
python - <<PY
import sys; sys.path.insert(0,".")
from famsa_amd import seqio
for n in (10000, 100000):
    c,o=seqio.synth_uniform(n,400); seqio.to_fasta(c,o,"/tmp/u_%d.fasta" % n)
PY
for n in 10000 100000; do
for cfg in "LCSGPU_UPGMA_CHAIN=0" "LCSGPU_UPGMA_CHAIN=1 LCSGPU_UPGMA_CHAIN_WG=16" "LCSGPU_UPGMA_CHAIN=1 LCSGPU_UPGMA_CHAIN_WG=32" "LCSGPU_UPGMA_CHAIN=0 LCSGPU_UPGMA_LAYOUT=triangle"; do
  for rep in 1 2; do
   echo "n=$n $cfg: $(env $cfg LCSGPU_PROFILE=1 timeout 120 famsa_amd/famsa-gpu -gt upgma -gt_export /tmp/u_$n.fasta /tmp/u.dnd 2>&1 | grep lcsgpu_upgma)"
  done
done
done
