#!/bin/bash
# round 5, GPU call H (final code): the suite with every duration, C5 with and without the LCS launches' LDS share, the
# end-to-end table, the bench's rocprof + PMC summaries, the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=0 ) > gpurun_out/h_suite.txt 2>&1
tail -8 gpurun_out/h_suite.txt
F=/tmp/family_3000000_300.fasta
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
seqio.family_fasta(3000000, 300, "$F")
PY
: > gpurun_out/h_c5_share.txt
for rep in 1 2 3 4 5; do
  for share in 41472 0; do
    LCSGPU_TUNE=lcs_share_lds=$share famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> /tmp/sw.err
    echo "share=$share $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/h_c5_share.txt
  done
done
cat gpurun_out/h_c5_share.txt
E2E_REFERENCE_FROM=profiles/e2e_r04.json timeout 1500 python scripts/e2e_compare.py r05 > gpurun_out/h_e2e.txt 2>&1
tail -20 gpurun_out/h_e2e.txt | cut -c1-300
bash scripts/profile_round.sh r05 > gpurun_out/h_profile_round.txt 2>&1
head -20 gpurun_out/rocprof_r05_summary.txt | cut -c1-150
python bench.py --pmc > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err
cat gpurun_out/bench_r05.json
