#!/bin/bash
# round 5, GPU call N: hardware queues for the lanes' streams (GPU_MAX_HW_QUEUES; the library's default is 16) with the
# one-workgroup searches, C5 at 3 000 000 sequences, three rounds alternating
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F=/tmp/family_3000000_300.fasta
python - <<PY
import sys
sys.path.insert(0, '.')
from famsa_amd import seqio
seqio.family_fasta(3000000, 300, "$F")
PY
: > gpurun_out/n_c5_hwq.txt
for rep in 1 2 3; do
  for q in 16 8 12 20 24; do
    GPU_MAX_HW_QUEUES=$q timeout 120 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> /tmp/sw.err
    echo "queues=$q $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/n_c5_hwq.txt
  done
done
sort -V gpurun_out/n_c5_hwq.txt
