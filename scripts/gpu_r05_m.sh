#!/bin/bash
# round 5, GPU call M: the last knob sweep of C5 on the final code (pool threads, batches, slice), three rounds, alternating
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F=/tmp/family_3000000_300.fasta
python - <<PY
import sys
sys.path.insert(0, '.')
from famsa_amd import seqio
seqio.family_fasta(3000000, 300, "$F")
PY
: > gpurun_out/m_c5_sweep.txt
run() { # label, env...
  label=$1; shift
  env "$@" timeout 120 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> /tmp/sw.err
  echo "$label $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/m_c5_sweep.txt
}
for rep in 1 2 3; do
  run "default" X=1
  run "pool=36" FAMSA_HOST_TEST=pool=36
  run "pool=40" FAMSA_HOST_TEST=pool=40
  run "pool=44" FAMSA_HOST_TEST=pool=44
  run "groups=3" LCSGPU_TUNE=clarans_groups=3
  run "groups=6" LCSGPU_TUNE=clarans_groups=6
  run "slice=700" LCSGPU_TUNE=clarans_slice_us=700
  run "slice=1500" LCSGPU_TUNE=clarans_slice_us=1500
  run "pool=40,groups=5" FAMSA_HOST_TEST=pool=40 LCSGPU_TUNE=clarans_groups=5
done
sort gpurun_out/m_c5_sweep.txt
