#!/bin/bash
# dev tool (GPU box; works on the scratch copy): the same run with famsa_amd/_old/liblcsgpu_old.so (a build of an
# earlier commit) in place of the library -> Newick hashes
cd "$(dirname "$0")/.."
python - <<'PY'
import sys
sys.path.insert(0, ".")
from famsa_amd import seqio
c, o = seqio.synth_uniform(100000, 400)
seqio.to_fasta(c, o, "/tmp/u.fasta")
PY
cp famsa_amd/liblcsgpu.so /tmp/new_lib.so; cp famsa_amd/_old/liblcsgpu_old.so famsa_amd/liblcsgpu.so
for gt in upgma upgma_modified; do
  timeout 200 famsa_amd/famsa-gpu -v -gt $gt -gt_export /tmp/u.fasta /tmp/u.$gt.dnd 2>&1 | tail -3
  sha256sum /tmp/u.$gt.dnd
done
cp /tmp/new_lib.so famsa_amd/liblcsgpu.so
