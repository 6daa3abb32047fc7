#!/bin/bash
# round 5, GPU call F: a whole CLARANS search in one workgroup (clarans_form=1) against the rounds (clarans_form=0)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_clarans.py -x -q -m gpu > gpurun_out/f_clarans_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/f_clarans_tests.txt
tail -15 gpurun_out/f_clarans_tests.txt
grep -q "rc=0" gpurun_out/f_clarans_tests.txt || exit 0
timeout 600 python -m pytest tests/test_gpu_atsize.py -x -q -m gpu -k "c5" > gpurun_out/f_c5_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/f_c5_tests.txt
tail -5 gpurun_out/f_c5_tests.txt
F=/tmp/family_3000000_300.fasta
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
seqio.family_fasta(3000000, 300, "$F")
seqio.family_fasta(1000000, 300, "/tmp/family_1000000_300.fasta")
PY
: > gpurun_out/f_c5_sweep.txt
run() { # label, file, env...
  label=$1; shift; file=$1; shift
  env "$@" timeout 120 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $file /tmp/sw.dnd 2> /tmp/sw.err
  echo "$label $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/f_c5_sweep.txt
}
for rep in 1 2 3; do
  run "3M search" $F X=1
  run "3M rounds" $F LCSGPU_TUNE=clarans_form=0
  run "3M search,share=0" $F LCSGPU_TUNE=lcs_share_lds=0
  run "3M search,slice=300" $F LCSGPU_TUNE=clarans_slice_us=300
  run "3M search,slice=3000" $F LCSGPU_TUNE=clarans_slice_us=3000
  run "3M search,groups=8" $F LCSGPU_TUNE=clarans_groups=8
  run "3M search,groups=2" $F LCSGPU_TUNE=clarans_groups=2
  run "3M search,pool=48" $F FAMSA_HOST_TEST=pool=48
  run "1M search" /tmp/family_1000000_300.fasta X=1
  run "1M rounds" /tmp/family_1000000_300.fasta LCSGPU_TUNE=clarans_form=0
done
cat gpurun_out/f_c5_sweep.txt
LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> gpurun_out/f_c5_profile.txt
grep -iE "clarans|tree_build|engine\." gpurun_out/f_c5_profile.txt | head -30
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o c5 -- /root/repo/famsa_amd/famsa-gpu -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd > /dev/null 2>&1
find /tmp/prof_f -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-200' > /root/repo/gpurun_out/f_c5_kernel_stats.txt
cat /root/repo/gpurun_out/f_c5_kernel_stats.txt
