#!/bin/bash
# round 5, GPU call E: two CLARANS looks in flight (C5 at 3 000 000 sequences against one look), sl next to slink at
# 100 000 sequences with the stage statistics, the whole suite with durations
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_clarans.py -x -q -m gpu > gpurun_out/e_clarans_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/e_clarans_tests.txt
tail -4 gpurun_out/e_clarans_tests.txt
F=/tmp/family_3000000_300.fasta
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
if not os.path.exists("$F"):
    seqio.family_fasta(3000000, 300, "$F")
seqio.family_fasta(1000000, 300, "/tmp/family_1000000_300.fasta")
PY
: > gpurun_out/e_c5_sweep.txt
run() { # label, file, env...
  label=$1; shift; file=$1; shift
  env "$@" famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $file /tmp/sw.dnd 2> /tmp/sw.err
  echo "$label $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/e_c5_sweep.txt
}
for rep in 1 2 3 4; do
  run "3M depth=2" $F X=1
  run "3M depth=1" $F LCSGPU_TUNE=clarans_depth=1
  run "3M depth=2,look=8" $F LCSGPU_TUNE=clarans_look=8
  run "3M depth=2,look=32" $F LCSGPU_TUNE=clarans_look=32
  run "3M depth=2,groups=2" $F LCSGPU_TUNE=clarans_groups=2
  run "3M depth=2,groups=8" $F LCSGPU_TUNE=clarans_groups=8
  run "3M depth=2,share=0" $F LCSGPU_TUNE=lcs_share_lds=0
  run "1M depth=2" /tmp/family_1000000_300.fasta X=1
  run "1M depth=1" /tmp/family_1000000_300.fasta LCSGPU_TUNE=clarans_depth=1
done
cat gpurun_out/e_c5_sweep.txt
LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> gpurun_out/e_c5_profile.txt
grep -iE "clarans|tree_build|look" gpurun_out/e_c5_profile.txt | head -30
# sl next to slink at C4's size (e2e_r05.json: 1.63 s against 1.39 s)
python - <<PY
import sys
sys.path.insert(0, '.')
from famsa_amd import seqio
codes, offsets = seqio.synth_uniform(100000, 400)
seqio.to_fasta(codes, offsets, "/tmp/synth100k.fasta")
PY
ls -la /tmp/synth100k.fasta
: > gpurun_out/e_sl_slink.txt
for rep in 1 2 3; do
  for gt in sl slink; do
    LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -gt $gt -gt_export /tmp/synth100k.fasta /tmp/o_$gt.dnd 2> /tmp/o.err
    echo "== $gt rep $rep" >> gpurun_out/e_sl_slink.txt
    grep -E "time\.|mst|prim|slink|order|gpu\." /tmp/o.err >> gpurun_out/e_sl_slink.txt
  done
done
tail -60 gpurun_out/e_sl_slink.txt
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=30 ) > gpurun_out/e_suite.txt 2>&1
tail -45 gpurun_out/e_suite.txt
