cd $GRAFT_REPO_ROOT
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
seqio.family_fasta(3000000, 300, "/tmp/f3m.fasta")
PY
for hook in leaves_last none; do
for rep in 1 2; do
echo "== $hook"
FAMSA_HOST_TEST=$hook LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/f3m.fasta /tmp/o.dnd 2>&1 | grep -E "fasttree.level [0-9]|fasttree.tail|tree_build|triangles_batch:|triangle batches|engine.tri"
done
done
