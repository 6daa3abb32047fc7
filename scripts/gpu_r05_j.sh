#!/bin/bash
# round 5, GPU call J: -dist_export with its rectangles' columns in length order against input order (LCS kernel time of the
# whole command; the sets are exported as they were read): the real 13 774-record set, hemopexin, a 30 000-member family set
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - <<PY
import sys, os
sys.path.insert(0, '.')
import numpy as np
from famsa_amd import seqio
seqio.realmix_fasta("tests/golden", "/tmp/realmix.fasta")
codes, offsets = seqio.family_set(30000, 300)
rng = np.random.default_rng(5)
codes, offsets = seqio.reorder(codes, offsets, rng.permutation(30000))
seqio.to_fasta(codes, offsets, "/tmp/family30k_shuffled.fasta")
PY
: > gpurun_out/j_dist_export.txt
for f in /tmp/realmix.fasta tests/golden/hemopexin/hemopexin /tmp/family30k_shuffled.fasta; do
  for rep in 1 2 3; do
    for hook in none csv_input_order; do
      t0=$(date +%s.%N)
      FAMSA_HOST_TEST=$hook famsa_amd/famsa-gpu -v -dist_export $f /tmp/o_$hook.csv 2> /tmp/o.err
      t1=$(date +%s.%N)
      echo "$(basename $f) columns=$([ $hook = none ] && echo by-length || echo input-order) $(grep -E 'gpu.lcs_kernel_ms|time.tree_build' /tmp/o.err | tr '\n' ' ') wall=$(python -c "print(round($t1-$t0,3))") sha=$(sha256sum /tmp/o_$hook.csv | cut -c1-12)" >> gpurun_out/j_dist_export.txt
    done
  done
done
cat gpurun_out/j_dist_export.txt
