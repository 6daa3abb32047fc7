#!/usr/bin/env python3
"""Rate of the long-ref kernel (refs > 2048 residues): n sequences of one length (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, famsa_amd
from famsa_amd import seqio
for L in (2100, 3000, 4200, 6000):
    n = 6000
    codes, offsets = seqio.synth_uniform(n, L, seed=5)
    eng = famsa_amd.LcsGpu(0); eng.upload(codes, offsets)
    pairs = n * (n - 1) // 2
    out = torch.empty(pairs, dtype=torch.int16, device="cuda:0")
    for r in range(2):
        eng.lcs_triangle_dev(0, n, out.data_ptr(), 2, sync=True)
        ms, nl = eng.last_kernel_ms()
    print(f"L={L}: {ms:.1f} ms  {pairs*L*L/(ms*1e-3)/1e12:.0f} Tcell/s  checksum {int(out.to(torch.int64).sum().item())}")
    eng.close()
