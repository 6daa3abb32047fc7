cd /root/repo
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
if not os.path.exists("/tmp/f3m.fasta"):
    seqio.family_fasta(3000000, 300, "/tmp/f3m.fasta")
PY
ROOT=/root/repo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5; FAMSA_GPU_CLEAN_EXIT=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o run -- $ROOT/famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export /tmp/f3m.fasta /tmp/o.dnd > /tmp/c5_trace_out.txt 2>&1; grep -E "tree_build|lcs_kernel" /tmp/c5_trace_out.txt; grep -iE "rocprof|error|warn" /tmp/c5_trace_out.txt | head -5; ls /tmp/prof_c5 2>&1 | head
python $ROOT/scripts/rocpd_summary.py $(find /tmp/prof_c5 -name "*.db") | head -45 > $ROOT/gpurun_out/c5_kernels_r06.txt
python $ROOT/scripts/rocpd_launches.py $(find /tmp/prof_c5 -name "*.db") "" 1500 > $ROOT/gpurun_out/c5_launches_r06.txt
cat $ROOT/gpurun_out/c5_kernels_r06.txt
