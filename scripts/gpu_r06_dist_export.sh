#!/bin/bash
# round 6: -dist_export with the text made on the device (lcsgpu_dist_text_*), blocks written by a pwrite team.
# Sets: hemopexin, C3 (10 000 x 400 aa), the real 13 774-record set, a shuffled 30 000-member family set.
# Output: gpurun_out/dist_export_r06.txt (one line per run; sha of the CSV; LCSGPU_PROFILE lines of one run per set)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=${OUT_DIR:-/tmp}
python - <<PY
import sys
sys.path.insert(0, '.')
import numpy as np
from famsa_amd import seqio
seqio.realmix_fasta("tests/golden", "/tmp/realmix.fasta")
codes, offsets = seqio.family_set(30000, 300)
rng = np.random.default_rng(5)
codes, offsets = seqio.reorder(codes, offsets, rng.permutation(30000))
seqio.to_fasta(codes, offsets, "/tmp/family30k_shuffled.fasta")
codes, offsets = seqio.synth_uniform(10000, 400)
seqio.to_fasta(codes, offsets, "/tmp/synth10k.fasta")
PY
R=gpurun_out/dist_export_r06.txt
: > $R
df -h $OUT | tail -1 >> $R
nproc >> $R
for f in tests/golden/hemopexin/hemopexin /tmp/synth10k.fasta /tmp/realmix.fasta /tmp/family30k_shuffled.fasta; do
  for hook in ${HOOKS:-none csv_host_format}; do
    for rep in 1 2 3; do
      rm -f $OUT/o_$hook.csv
      t0=$(date +%s.%N)
      FAMSA_HOST_TEST=$hook famsa_amd/famsa-gpu -v -dist_export $EXTRA $f $OUT/o_$hook.csv 2> /tmp/o.err
      t1=$(date +%s.%N)
      echo "$(basename $f) formatter=$([ $hook = csv_host_format ] && echo host || echo device:$hook) $(grep -E 'gpu.lcs_kernel_ms|time.tree_build' /tmp/o.err | tr '\n' ' ') wall=$(python -c "print(round($t1-$t0,3))") bytes=$(stat -c %s $OUT/o_$hook.csv) sha=$(sha256sum $OUT/o_$hook.csv | cut -c1-12)" >> $R
    done
  done
  LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -dist_export $f $OUT/o_p.csv 2>&1 | grep -iE "text|dist_export|time\.|create" | sed "s/^/  [$(basename $f)] /" >> $R
done
cat $R
