#!/bin/bash
# dev tool (GPU box): resident host memory of famsa-gpu along its stages for a small plain run, a small MedoidTree run and a
# large one -> gpurun_out/rss_probe.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - <<'PY'
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
if not os.path.exists("/tmp/u_10000.fasta"):
    c, o = seqio.synth_uniform(10000, 400); seqio.to_fasta(c, o, "/tmp/u_10000.fasta")
for n in (200000, 3000000):
    f = "/tmp/fam_%d.fasta" % n
    if not os.path.exists(f):
        seqio.family_fasta(n, 300, f)
PY
OUT=gpurun_out/rss_probe.txt
: > $OUT
probe() { # label, args...
  label=$1; shift
  sleep 2
  timeout 300 famsa_amd/famsa-gpu -v "$@" /tmp/rss.dnd 2> /tmp/rss.err
  echo "$label: $(grep -E 'mem\.|time.tree_build' /tmp/rss.err | tr '\n' ' ')" >> $OUT
}
probe "10k sl" -gt sl -gt_export /tmp/u_10000.fasta
probe "200k medoid upgma" -medoidtree -gt upgma -gt_export /tmp/fam_200000.fasta
FAMSA_HOST_TEST=pool=8 probe "200k medoid upgma, 8 pool threads" -medoidtree -gt upgma -gt_export /tmp/fam_200000.fasta
probe "3M medoid upgma" -medoidtree -gt upgma -gt_export /tmp/fam_3000000.fasta
FAMSA_HOST_TEST=pool=8 probe "3M medoid upgma, 8 pool threads" -medoidtree -gt upgma -gt_export /tmp/fam_3000000.fasta
cat $OUT
