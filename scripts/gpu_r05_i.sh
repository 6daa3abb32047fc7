#!/bin/bash
# round 5, GPU call I: the suite on the final code, five plain C5 runs, three at 10^6
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 ) > gpurun_out/i_suite.txt 2>&1
tail -28 gpurun_out/i_suite.txt
python - <<PY
import sys
sys.path.insert(0, '.')
from famsa_amd import seqio
seqio.family_fasta(3000000, 300, "/tmp/family_3000000_300.fasta")
seqio.family_fasta(1000000, 300, "/tmp/family_1000000_300.fasta")
PY
: > gpurun_out/i_c5_runs.txt
for rep in 1 2 3 4 5; do
  famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/family_3000000_300.fasta /tmp/sw.dnd 2> /tmp/sw.err
  echo "3M run $rep $(grep -E 'time.tree_build|gpu.lcs_kernel_ms|time.main_until_exit' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/i_c5_runs.txt
done
for rep in 1 2 3; do
  famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/family_1000000_300.fasta /tmp/sw.dnd 2> /tmp/sw.err
  echo "1M run $rep $(grep -E 'time.tree_build|gpu.lcs_kernel_ms|time.main_until_exit' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/i_c5_runs.txt
done
cat gpurun_out/i_c5_runs.txt
bash scripts/c5_profile.sh > /dev/null 2>&1; cp gpurun_out/c5_profile.txt gpurun_out/i_c5_profile.txt
grep -E "tree_build|clarans.calls|sha256" gpurun_out/i_c5_profile.txt
