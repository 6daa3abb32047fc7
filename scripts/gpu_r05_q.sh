#!/bin/bash
# round 5, GPU call Q: the final pool policy -- the GPU suite, C5 plain runs at 3 x 10^6 / 10^6 / 200 000, the C5 profile
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=5 ) > gpurun_out/q_suite.txt 2>&1
tail -14 gpurun_out/q_suite.txt
python - <<PY
import sys
sys.path.insert(0, '.')
from famsa_amd import seqio
seqio.family_fasta(3000000, 300, "/tmp/family_3000000_300.fasta")
seqio.family_fasta(1000000, 300, "/tmp/family_1000000_300.fasta")
seqio.family_fasta(200000, 300, "/tmp/family_200000_300.fasta")
PY
: > gpurun_out/q_c5_runs.txt
for n in 3000000 3000000 3000000 3000000 3000000 1000000 1000000 1000000 200000 200000 200000; do
  famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/family_${n}_300.fasta /tmp/sw.dnd 2> /tmp/sw.err
  echo "n=$n $(grep -E 'time.tree_build|gpu.lcs_kernel_ms|time.main_until_exit' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/q_c5_runs.txt
done
cat gpurun_out/q_c5_runs.txt
bash scripts/c5_profile.sh > /dev/null 2>&1; cp gpurun_out/c5_profile.txt gpurun_out/q_c5_profile.txt
grep -E "tree_build|clarans.calls|sha256" gpurun_out/q_c5_profile.txt
