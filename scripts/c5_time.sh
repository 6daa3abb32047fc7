#!/bin/bash
# dev tool (GPU box): -medoidtree -gt upgma tree stage at 1 000 000 and 3 000 000 family sequences, leaves on the device
# (FAMSA_LEAF_DEVICE=1) and on the host (default) -> gpurun_out/c5_time.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - <<'PY'
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
for n in (1000000, 3000000):
    f = "/tmp/fam_%d.fasta" % n
    if not os.path.exists(f):
        seqio.family_fasta(n, 300, f)
PY
: > gpurun_out/c5_time.txt
for n in 1000000 3000000; do
  for leaf in device host; do
    for rep in 1 2; do
      if [ $leaf = device ]; then export FAMSA_LEAF_DEVICE=1; else unset FAMSA_LEAF_DEVICE; fi
      FAMSA_GPU_PROFILE=1 timeout 300 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/fam_$n.fasta /tmp/fam_$n.dnd 2> /tmp/c5.err
      echo "n=$n leaves=$leaf rc=$? $(grep -E 'time.tree_build|fasttree.partial_trees|fasttree.clarans|fasttree.lcs_calls' /tmp/c5.err | tr '\n' ' ') sha=$(sha256sum /tmp/fam_$n.dnd | cut -c1-16)" >> gpurun_out/c5_time.txt
    done
  done
done
unset FAMSA_LEAF_DEVICE
cat gpurun_out/c5_time.txt
