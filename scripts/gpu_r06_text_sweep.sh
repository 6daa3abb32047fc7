cd $GRAFT_REPO_ROOT
python - <<PY
import sys
sys.path.insert(0, '.')
import numpy as np
from famsa_amd import seqio
codes, offsets = seqio.family_set(30000, 300)
rng = np.random.default_rng(5)
codes, offsets = seqio.reorder(codes, offsets, rng.permutation(30000))
seqio.to_fasta(codes, offsets, "/tmp/family30k_shuffled.fasta")
codes, offsets = seqio.synth_uniform(10000, 400)
seqio.to_fasta(codes, offsets, "/tmp/synth10k.fasta")
PY
LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -dist_export /tmp/synth10k.fasta /tmp/o.csv 2>&1 | grep -E "dist_text block|dist_export.text" | tail -8
for f in /tmp/synth10k.fasta /tmp/family30k_shuffled.fasta; do
for cfg in "text_writers=2" "text_writers=4" "text_writers=8" "text_writers=16" "text_writers=32" "text_writers=16,text_block_mb=8" "text_writers=16,text_block_mb=16" "text_writers=16,text_block_mb=64" "text_writers=16,text_slots=2" "text_writers=16,text_slots=4" "text_writers=16,text_slots=6"; do
  for rep in 1 2; do
  rm -f /tmp/o.csv
  FAMSA_HOST_TEST=$cfg famsa_amd/famsa-gpu -v -dist_export $f /tmp/o.csv 2>&1 | grep -E "time.tree_build" | sed "s|^|$(basename $f) $cfg |"
  done
done
done
