#!/bin/bash
# round 6 (second session): rows of D in flight per batch of the CLARANS chain's flags phase (CLARANS_QC: 4 shipped, 8, 16 as
# library variants under famsa_amd/_variants), then -gt upgma at 100 000 x 400 aa right after the 3 x 10^6 runs (its slow mode)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
R=gpurun_out/qc_r06.txt
: > $R
python - <<PY
import sys, os
sys.path.insert(0, ".")
from famsa_amd import seqio
for n in (1000000, 3000000):
    p = "/tmp/family_%d_300.fasta" % n
    if not os.path.exists(p):
        seqio.family_fasta(n, 300, p)
codes, offsets = seqio.synth_uniform(100000, 400)
seqio.to_fasta(codes, offsets, "/tmp/synth100k.fasta")
PY
for v in base qc8 qc16; do
  LIB=famsa_amd/_variants/$v/liblcsgpu.so; [ $v = base ] && LIB=famsa_amd/liblcsgpu.so
  echo "== $v: one chain (2000 members, 100 medoids)" >> $R
  LCSGPU_LIB=$LIB python scripts/clarans_bench.py 6 2>&1 | tail -4 >> $R
done
for rep in 1 2 3; do
for v in base qc8 qc16; do
  LIBDIR=famsa_amd/_variants/$v; [ $v = base ] && LIBDIR=famsa_amd
  for n in 1000000 3000000; do
    WANT=$(python -c "import json; print(json.load(open('tests/golden/meta_large.json')).get('family$n', {}).get('medoid_upgma_newick_sha256', 'no-pin'))")
    LD_LIBRARY_PATH=$LIBDIR famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/family_${n}_300.fasta /tmp/o.dnd 2> /tmp/o.err
    echo "family$n $v $(grep -E 'time.tree_build' /tmp/o.err | tr '\n' ' ') newick=$([ "$(sha256sum /tmp/o.dnd | cut -d' ' -f1)" = "$WANT" ] && echo identical || echo DIFFERENT)" >> $R
  done
done
done
for rep in 1 2 3; do
  echo "== upgma 100 000 x 400 aa, run $rep right after" >> $R
  LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -gt upgma -gt_export /tmp/synth100k.fasta /tmp/u.dnd 2>&1 | grep -E "lcsgpu_upgma: [0-9.]+ s|allocated in|distances \+ row|time.tree_build" | cut -c1-200 >> $R
done
cat $R
