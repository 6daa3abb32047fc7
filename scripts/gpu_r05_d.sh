#!/bin/bash
# round 5, GPU call D: the CLARANS distance matrix as a full square, fewer step workgroups / one LCS stream / a polling
# driver as knobs (C5 at 3 000 000 sequences), the whole suite with durations, the bench's rocprof + PMC summaries, the
# end-to-end table
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_clarans.py -x -q -m gpu > gpurun_out/d_clarans_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/d_clarans_tests.txt
tail -4 gpurun_out/d_clarans_tests.txt
F=/tmp/family_3000000_300.fasta
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
if not os.path.exists("$F"):
    seqio.family_fasta(3000000, 300, "$F")
PY
: > gpurun_out/d_c5_sweep.txt
run() { # label, env...
  label=$1; shift
  env "$@" famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> /tmp/sw.err
  echo "$label $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/d_c5_sweep.txt
}
for rep in 1 2 3; do
  run "default" X=1
  run "wgs=8" LCSGPU_TUNE=clarans_wgs=8
  run "wgs=4" LCSGPU_TUNE=clarans_wgs=4
  run "serial" LCSGPU_TUNE=lcs_serial=1
  run "serial,share=0" LCSGPU_TUNE=lcs_serial=1,lcs_share_lds=0
  run "spin" LCSGPU_TUNE=clarans_spin=1
  run "wgs=8,spin" LCSGPU_TUNE=clarans_wgs=8,clarans_spin=1
  run "wgs=8,pool=48" LCSGPU_TUNE=clarans_wgs=8 FAMSA_HOST_TEST=pool=48
  run "wgs=8,pool=64" LCSGPU_TUNE=clarans_wgs=8 FAMSA_HOST_TEST=pool=64
  run "groups=3" LCSGPU_TUNE=clarans_groups=3
  run "groups=2" LCSGPU_TUNE=clarans_groups=2
  run "wgs=8,serial,spin" LCSGPU_TUNE=clarans_wgs=8,lcs_serial=1,clarans_spin=1
done
cat gpurun_out/d_c5_sweep.txt
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=40 ) > gpurun_out/d_suite.txt 2>&1
tail -50 gpurun_out/d_suite.txt
bash scripts/profile_round.sh r05 > gpurun_out/d_profile_round.txt 2>&1
head -30 gpurun_out/rocprof_r05_summary.txt | cut -c1-150
E2E_REFERENCE_FROM=profiles/e2e_r04.json timeout 1500 python scripts/e2e_compare.py r05 > gpurun_out/d_e2e.txt 2>&1
tail -20 gpurun_out/d_e2e.txt | cut -c1-300
