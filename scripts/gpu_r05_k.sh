#!/bin/bash
# round 5, GPU call K (run at commit 9d9b1bc, which was reverted: clarans_depth exists only there): two launches of a CLARANS batch in
# flight (clarans_depth=2) against one, slices of 250 / 500 / 1000 us -> profiles/c5_looks_r05.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_clarans.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_atsize.py tests/test_gpu_realmix.py -x -q -m gpu -k "c5 or medoid" 2>&1 | tail -3
F=/tmp/family_3000000_300.fasta
python - <<PY
import sys
sys.path.insert(0, '.')
from famsa_amd import seqio
seqio.family_fasta(3000000, 300, "$F")
PY
: > gpurun_out/k_c5_depth.txt
run() { # label, env...
  label=$1; shift
  env "$@" timeout 120 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> /tmp/sw.err
  echo "$label $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/k_c5_depth.txt
}
for rep in 1 2 3 4; do
  run "depth=2 slice=500" X=1
  run "depth=1 slice=1000" LCSGPU_TUNE=clarans_depth=1,clarans_slice_us=1000
  run "depth=2 slice=1000" LCSGPU_TUNE=clarans_slice_us=1000
  run "depth=2 slice=250" LCSGPU_TUNE=clarans_slice_us=250
  run "depth=1 slice=500" LCSGPU_TUNE=clarans_depth=1
done
cat gpurun_out/k_c5_depth.txt
LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> /tmp/sw2.err
grep -E "tree_build|clarans.calls|engine.clarans" /tmp/sw2.err
