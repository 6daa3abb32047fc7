#!/bin/bash
# dev tool (GPU box): kernel trace of -gt upgma at 10 000 / 100 000 x 400 aa -> gpurun_out/upgma_profile.txt
cd "$(dirname "$0")/.."
ROOT=$(pwd)
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
cd /tmp && export TMPDIR=/tmp
: > $ROOT/gpurun_out/upgma_profile.txt
for n in ${1:-10000 100000}; do
  rm -rf /tmp/uprof
  FAMSA_GPU_CLEAN_EXIT=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/uprof -o run -- $ROOT/famsa_amd/famsa-gpu -v -gt upgma -gt_export /tmp/u_$n.fasta /tmp/u.dnd > /tmp/uprof.log 2>&1
  echo "== n=$n $(grep tree_build /tmp/uprof.log)" >> $ROOT/gpurun_out/upgma_profile.txt
  python $ROOT/scripts/rocpd_summary.py $(find /tmp/uprof -name "*.db") | head -8 >> $ROOT/gpurun_out/upgma_profile.txt
  python $ROOT/scripts/rocpd_timeline.py $(find /tmp/uprof -name "*.db") upgma_step | tail -5 >> $ROOT/gpurun_out/upgma_profile.txt
done
cat $ROOT/gpurun_out/upgma_profile.txt
