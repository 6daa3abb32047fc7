#!/bin/bash
# round 6 (second session): -gt upgma at 100 000 x 400 aa takes 1.7 s on some runs and 2.6 s on others: which phase
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
R=gpurun_out/upgma_modes_r06.txt
: > $R
python - <<PY
import sys
sys.path.insert(0, ".")
from famsa_amd import seqio
codes, offsets = seqio.synth_uniform(100000, 400)
seqio.to_fasta(codes, offsets, "/tmp/synth100k.fasta")
PY
for rep in 1 2 3 4; do
  echo "== run $rep ($(date +%T))" >> $R
  LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -gt upgma -gt_export /tmp/synth100k.fasta /tmp/u.dnd 2>&1 | grep -E "lcsgpu_upgma|time.tree_build|time.gpu_|reserve|hipMalloc|upgma" | cut -c1-260 >> $R
  rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" >> $R
  if [ $rep = 2 ]; then sleep 20; fi
done
cat $R
