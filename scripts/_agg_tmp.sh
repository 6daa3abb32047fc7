F=/tmp/family_3000000_300.fasta
python - <<PY
import sys
sys.path.insert(0, '.')
from famsa_amd import seqio
seqio.family_fasta(3000000, 300, "$F")
seqio.family_fasta(1000000, 300, "/tmp/family_1000000_300.fasta")
PY
run() { # label, file, env...
  label=$1; shift; file=$1; shift
  env "$@" timeout 120 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $file /tmp/sw.dnd 2> /tmp/sw.err
  echo "$label $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)"
}
for rep in 1 2 3; do
run "3M search pool=32" $F X=1
run "3M rounds pool=32" $F LCSGPU_TUNE=clarans_form=0
run "3M search pool=40" $F FAMSA_HOST_TEST=pool=40
run "3M search pool=48" $F FAMSA_HOST_TEST=pool=48
run "3M search slice=500" $F LCSGPU_TUNE=clarans_slice_us=500
run "1M search" /tmp/family_1000000_300.fasta X=1
run "1M rounds" /tmp/family_1000000_300.fasta LCSGPU_TUNE=clarans_form=0
done
LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> /tmp/sw2.err
grep -E "tree stage|tree_build|fasttree\.|engine\." /tmp/sw2.err | head -40
timeout 900 python -m pytest tests/test_gpu_atsize.py tests/test_gpu_realmix.py tests/test_gpu_endtoend.py -x -q -m gpu -k "c5 or medoid or fast" 2>&1 | tail -5
