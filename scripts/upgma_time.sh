#!/bin/bash
# dev tool (GPU box): -gt upgma tree stage at 10 000 and 100 000 x 400 aa -> gpurun_out/upgma_time.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - <<'PY'
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
for n in (10000, 100000):
    f = "/tmp/u_%d.fasta" % n
    if not os.path.exists(f):
        c, o = seqio.synth_uniform(n, 400)
        seqio.to_fasta(c, o, f)
PY
: > gpurun_out/upgma_time.txt
for n in 10000 100000; do
  for gt in upgma upgma_modified; do
    timeout 300 famsa_amd/famsa-gpu -v -gt $gt -gt_export /tmp/u_$n.fasta /tmp/u_$n.$gt.dnd 2> /tmp/u.err
    echo "n=$n gt=$gt rc=$? $(grep -E 'tree_build|lcs_kernel' /tmp/u.err | tr '\n' ' ') sha=$(sha256sum /tmp/u_$n.$gt.dnd | cut -c1-64)" >> gpurun_out/upgma_time.txt
  done
done
cat gpurun_out/upgma_time.txt
