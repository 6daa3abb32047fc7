#!/bin/bash
# dev tool (GPU box): 3 x 10^6-sequence MedoidTree against the number of hardware queues -> gpurun_out/hwq_sweep.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
N=${1:-3000000}
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
f = "/tmp/fam_$N.fasta"
if not os.path.exists(f):
    seqio.family_fasta($N, 300, f)
PY
OUT=gpurun_out/hwq_sweep.txt
: > $OUT
TIMEFORMAT='wall=%R'
for rep in 1 2; do
  for q in ${QUEUES:-16 8 12}; do
    sleep 2
    { time GPU_MAX_HW_QUEUES=$q timeout 120 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/fam_$N.fasta /tmp/hwq.dnd 2> /tmp/hwq.err ; } 2> /tmp/hwq.time
    echo "queues=$q rep=$rep $(cat /tmp/hwq.time) $(grep -E 'time.tree_build|mem.after_upload|mem.VmRSS' /tmp/hwq.err | tr '\n' ' ') sha=$(sha256sum /tmp/hwq.dnd | cut -c1-12)" >> $OUT
  done
done
cat $OUT
