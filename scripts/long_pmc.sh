#!/bin/bash
# Dev tool (GPU box): PMC counters of the long-ref kernel next to the 64-half-word register-resident kernel.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/long_one.py <<'P'
import sys, os
sys.path.insert(0, os.environ["ROOT"])
import torch, famsa_amd
from famsa_amd import seqio
for L, n in ((3000, 4000), (2048, 5000)):
    codes, offsets = seqio.synth_uniform(n, L, seed=5)
    eng = famsa_amd.LcsGpu(0); eng.upload(codes, offsets)
    out = torch.empty(n * (n - 1) // 2, dtype=torch.int16, device="cuda:0")
    eng.lcs_triangle_dev(0, n, out.data_ptr(), 2, sync=True)
    ms, nl = eng.last_kernel_ms(); print(L, ms)
    eng.close()
P
for set in "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_CYCLES" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAVES_LT_64"; do
  rm -rf /tmp/prof_l; ROOT=$ROOT rocprofv3 --kernel-trace --pmc $set -d /tmp/prof_l -o run -- python /tmp/long_one.py > /dev/null 2>&1
  python $ROOT/scripts/rocpd_summary.py $(find /tmp/prof_l -name "*.db") | grep -E "lcs_long_kernel|lcs_rows_kernel_pipe<64" 
done
