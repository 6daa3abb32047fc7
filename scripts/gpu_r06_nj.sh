#!/bin/bash
# round 6: neighbour joining as one resident launch against four launches per merge (hemopexin, 4188 sequences),
# the launch's own account of its time, and the ordered-sum check with its timings on an idle chip
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/nj_r06.txt
: > $OUT
for tune in "nj_loop=1" "nj_loop=1,nj_squeeze_min=100000" "nj_loop=1,nj_groups=128" "nj_loop=1,nj_groups=64" "nj_loop=0"; do
  for i in 1 2 3; do
    line=$(LCSGPU_TUNE=$tune timeout 120 famsa_amd/famsa-gpu -v -gt nj -gt_export tests/golden/hemopexin/hemopexin /tmp/nj.dnd 2>&1 | grep -E "tree_build|rror" | tr '\n' ' ')
    echo "$tune $line sha=$(sha256sum /tmp/nj.dnd | cut -c1-12)" | tee -a $OUT
  done
done
LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -gt nj -gt_export tests/golden/hemopexin/hemopexin /tmp/nj.dnd 2>&1 | grep -E "lcsgpu_nj" | tee -a $OUT
echo "--- ordered_sum_check 240 (one wave per CU: an idle chip)" | tee -a $OUT
famsa_amd/_build/ordered_sum_check 240 | tee -a $OUT
echo "--- ordered_sum_check (6000 vectors)" | tee -a $OUT
famsa_amd/_build/ordered_sum_check | tail -2 | tee -a $OUT
