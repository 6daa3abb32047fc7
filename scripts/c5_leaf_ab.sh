#!/bin/bash
# dev tool (GPU box): -medoidtree -gt upgma at 3 000 000 family sequences, the host leaf UPGMA in its two forms
# (square matrix = default, FAMSA_UPGMA_TRIANGLE=1 = the triangle walk), alternating -> gpurun_out/c5_leaf_ab.txt
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
: > gpurun_out/c5_leaf_ab.txt
for rep in 1 2 3; do
  for form in triangle square; do
    if [ $form = triangle ]; then export FAMSA_UPGMA_TRIANGLE=1; else unset FAMSA_UPGMA_TRIANGLE; fi
    FAMSA_GPU_PROFILE=1 timeout 300 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/fam_$N.fasta /tmp/fam_$N.dnd 2> /tmp/c5.err
    echo "n=$N leaves=$form rc=$? $(grep -E 'time.tree_build|fasttree.partial_trees|fasttree.clarans|fasttree.lcs_calls' /tmp/c5.err | tr '\n' ' ') sha=$(sha256sum /tmp/fam_$N.dnd | cut -c1-16)" >> gpurun_out/c5_leaf_ab.txt
  done
done
unset FAMSA_UPGMA_TRIANGLE
cat gpurun_out/c5_leaf_ab.txt
