#!/bin/bash
# round 5, GPU call B: UPGMA with compaction + streamed build (tests, C4 timing), C5 with merged half-word classes and the
# LDS share of the FastTree launches (A/B against lcs_share_lds=0), the bench line, the whole suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_endtoend.py -x -q -m gpu -k "upgma" > gpurun_out/b_upgma_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/b_upgma_tests.txt
tail -5 gpurun_out/b_upgma_tests.txt
timeout 900 python -m pytest tests/test_gpu_atsize.py -x -q -m gpu -k "upgma" > gpurun_out/b_upgma_atsize.txt 2>&1; echo "rc=$?" >> gpurun_out/b_upgma_atsize.txt
tail -5 gpurun_out/b_upgma_atsize.txt
# C4 -gt upgma, twice, with the engine's own account of the stages
python - <<'PY'
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
f = "/tmp/synth100k.fasta"
if not os.path.exists(f):
    c, o = seqio.synth_uniform(100000, 400)
    seqio.to_fasta(c, o, f)
PY
: > gpurun_out/b_c4_upgma.txt
for rep in 1 2; do
  sleep 5
  LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -gt upgma -gt_export /tmp/synth100k.fasta /tmp/u.dnd 2>&1 | grep -E "lcsgpu_upgma|time\.|gpu\.|mem\." >> gpurun_out/b_c4_upgma.txt
  rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" >> gpurun_out/b_c4_upgma.txt
  echo "sha $(sha256sum /tmp/u.dnd | cut -c1-64)" >> gpurun_out/b_c4_upgma.txt
done
cat gpurun_out/b_c4_upgma.txt
# C5: default, then without the LDS share
bash scripts/c5_profile.sh 3000000 > /dev/null 2>&1
cp gpurun_out/c5_profile.txt gpurun_out/b_c5_default.txt
F=/tmp/family_3000000_300.fasta
: > gpurun_out/b_c5_ab.txt
for rep in 1 2; do
  for share in 41472 0; do
    LCSGPU_TUNE=lcs_share_lds=$share famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/ab.dnd 2> /tmp/ab.err
    echo "lcs_share_lds=$share $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/ab.err | tr '\n' ' ') sha=$(sha256sum /tmp/ab.dnd | cut -c1-16)" >> gpurun_out/b_c5_ab.txt
  done
done
cat gpurun_out/b_c5_ab.txt
bash scripts/clarans_profile.sh 3000000 > /dev/null 2>&1
cp gpurun_out/clarans_kernel_stats.txt gpurun_out/b_c5_kernels.txt
head -22 gpurun_out/b_c5_kernels.txt | cut -c1-140
grep -E "clarans_round_kernel<1>.*n=|by grid_y: 1:3|span|in flight" gpurun_out/b_c5_kernels.txt | cut -c1-220 | head -12
python bench.py --steps 5 --warmup 2 > gpurun_out/b_bench.txt 2>gpurun_out/b_bench.err; tail -1 gpurun_out/b_bench.txt | cut -c1-600
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/b_suite.txt 2>&1
tail -8 gpurun_out/b_suite.txt
