#!/bin/bash
# round 6: -medoidtree -gt upgma tree stage at 200 000 / 1 000 000 / 3 000 000 family sequences, three plain runs each
# (time.tree_build, sha256 of the Newick against the reference's pin) -> gpurun_out/c5_runs_r06.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - <<PY
import sys, os
sys.path.insert(0, ".")
from famsa_amd import seqio
for n in (200000, 1000000, 3000000):
    p = "/tmp/family_%d_300.fasta" % n
    if not os.path.exists(p):
        seqio.family_fasta(n, 300, p)
PY
R=gpurun_out/c5_runs_r06.txt
: > $R
for n in 200000 1000000 3000000; do
  WANT=$(python -c "import json; print(json.load(open('tests/golden/meta_large.json')).get('family$n', {}).get('medoid_upgma_newick_sha256', 'no-pin'))")
  for rep in 1 2 3; do
    t0=$(date +%s.%N)
    famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/family_${n}_300.fasta /tmp/o.dnd 2> /tmp/o.err
    t1=$(date +%s.%N)
    echo "family$n $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/o.err | tr '\n' ' ') wall=$(python -c "print(round($t1-$t0,3))") newick=$([ "$(sha256sum /tmp/o.dnd | cut -d' ' -f1)" = "$WANT" ] && echo identical-to-the-reference || echo DIFFERENT)" >> $R
  done
done
cat $R
