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
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_VMEM"; do
  rm -rf /tmp/prof_l; ROOT=$ROOT rocprofv3 --kernel-trace --pmc $set -d /tmp/prof_l -o run -- python /tmp/long_one.py > /dev/null 2>&1
  python $ROOT/scripts/rocpd_summary.py $(find /tmp/prof_l -name "*.db") | grep -E "lcs_long_kernel|lcs_rows_kernel_pipe<64" 
done
