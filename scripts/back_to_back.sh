#!/bin/bash
# dev tool (GPU box): does a run pay for the device memory the previous process has just left?  -gt upgma / -gt sl at
# 100 000 x 400 aa back to back and with a pause -> gpurun_out/back_to_back.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - <<'PY'
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
f = "/tmp/u_100000.fasta"
if not os.path.exists(f):
    c, o = seqio.synth_uniform(100000, 400)
    seqio.to_fasta(c, o, f)
PY
OUT=gpurun_out/back_to_back.txt
: > $OUT
run() { # gt label
  t0=$(date +%s.%N)
  LCSGPU_PROFILE=1 timeout 300 famsa_amd/famsa-gpu -v -gt $1 -gt_export /tmp/u_100000.fasta /tmp/u.dnd 2> /tmp/u.err
  rc=$?
  echo "$2 gt=$1 rc=$rc wall=$(python3 -c "import time,sys; print('%.2f' % (time.time() - float(sys.argv[1])))" $t0) $(grep -E 'time.tree_build|main_until_exit' /tmp/u.err | tr '\n' ' ')" >> $OUT
}
sleep 5
run upgma "after 5 s of rest:   "
run upgma "right after a upgma: "
sleep 5
run upgma "after 5 s of rest:   "
sleep 5
run sl "after 5 s of rest:   "
run sl "right after a sl:    "
run upgma "right after a sl:    "
sleep 5
export FAMSA_GPU_CLEAN_EXIT=1
run upgma "clean exit, after rest:  "
run upgma "clean exit, right after: "
unset FAMSA_GPU_CLEAN_EXIT
cat $OUT
