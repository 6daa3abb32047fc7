python - <<'PY'
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
seqio.family_fasta(200000, 300, "/tmp/fam200k.fasta")
PY
LCSGPU_CLARANS_CHAIN_DBG=1 LCSGPU_PROFILE=1 LCSGPU_CLARANS_CHAIN=1 FAMSA_GPU_PROFILE=1 timeout 300 famsa_amd/famsa-gpu -v -t 1 -medoidtree -gt upgma -gt_export /tmp/fam200k.fasta /tmp/o.dnd 2> /tmp/c5.err
grep -E "clarans.chain |time.tree_build" /tmp/c5.err | head -12
