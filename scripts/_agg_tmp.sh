timeout 900 python -m pytest tests/test_gpu_clarans.py -x -q -m gpu 2>&1 | tail -5
F=/tmp/family_3000000_300.fasta
python - <<PY
import sys
sys.path.insert(0, '.')
from famsa_amd import seqio
seqio.family_fasta(3000000, 300, "$F")
PY
python scripts/clarans_bench.py 3 2>&1 | tail -3
run() { # label, env...
  label=$1; shift
  env "$@" timeout 120 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> /tmp/sw.err
  echo "$label $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)"
}
for rep in 1 2 3; do
run "search pool=32" X=1
run "rounds pool=32" LCSGPU_TUNE=clarans_form=0
run "search pool=48" FAMSA_HOST_TEST=pool=48
run "search pool=64" FAMSA_HOST_TEST=pool=64
run "search share=0" LCSGPU_TUNE=lcs_share_lds=0
run "search slice=500" LCSGPU_TUNE=clarans_slice_us=500
run "search slice=2000" LCSGPU_TUNE=clarans_slice_us=2000
done
LCSGPU_PROFILE=1 famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> /tmp/sw2.err
grep -E "tree stage|tree_build|fasttree\.|engine\.|clarans\." /tmp/sw2.err | head -60
