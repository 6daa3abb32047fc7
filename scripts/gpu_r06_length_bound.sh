#!/bin/bash
# round 6: the length bound of the MST rounds that recompute their LCS values (FuseArgs::prune) -- on / off, on a set of
# four families of very different lengths (200 000 sequences), on the ragged 100 000-member family set (210-300 aa: its
# lengths are too close for the bound to cut anything) and on the uniform 100 000 x 400 aa set (the test's cost alone)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - <<PY
import sys
sys.path.insert(0, ".")
import numpy as np
from famsa_amd import seqio
rng = np.random.Generator(np.random.PCG64(23))
fams = []
for length, members in ((60, 50000), (150, 50000), (420, 50000), (900, 50000)):
    anc = rng.integers(0, 20, size=length, dtype=np.uint8)
    S = np.tile(anc, (members, 1))
    m = rng.random(S.shape) < 0.2
    S[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
    lens = rng.integers(int(length * 0.9), length + 1, size=members)
    fams += [S[i, : lens[i]].copy() for i in range(members)]
order = rng.permutation(len(fams))
codes, offsets = seqio.pack([fams[i] for i in order])
seqio.to_fasta(codes, offsets, "/tmp/four_families_200k.fasta")
seqio.family_fasta(100000, 300, "/tmp/family100k.fasta")
codes, offsets = seqio.synth_uniform(100000, 400)
seqio.to_fasta(codes, offsets, "/tmp/synth100k.fasta")
PY
R=gpurun_out/length_bound_r06.txt
: > $R
for f in /tmp/four_families_200k.fasta /tmp/family100k.fasta /tmp/synth100k.fasta; do
  for mode in "passes:" "recompute:" "recompute:mst_length_bound=0"; do
    m=${mode%%:*}; t=${mode#*:}
    for rep in 1 2; do
      LCSGPU_MST_MODE=$m LCSGPU_TUNE=$t famsa_amd/famsa-gpu -vv -gt sl -gt_export $f /tmp/o.dnd 2> /tmp/o.err
      echo "$(basename $f) mode=$m ${t:-bound=on} $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/o.err | tr '\n' ' ') $(grep -o '[0-9]* computed, [0-9]* let go[^)]*)' /tmp/o.err) sha=$(sha256sum /tmp/o.dnd | cut -c1-12)" >> $R
    done
  done
done
cat $R
