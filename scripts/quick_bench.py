#!/usr/bin/env python3
"""Quick single-GPU timing of the triangle kernel on the synthetic uniform set (dev tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import famsa_amd
from famsa_amd import seqio

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 400
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
codes, offsets = seqio.synth_uniform(n, L)
eng = famsa_amd.LcsGpu(0)
t0 = time.time(); eng.upload(codes, offsets); print("upload s", time.time() - t0)
pairs = n * (n - 1) // 2
out = torch.empty(pairs, dtype=torch.int16, device="cuda:0")
for r in range(reps):
    t0 = time.time()
    eng.lcs_triangle_dev(0, n, out.data_ptr(), 2, sync=True)
    dt = time.time() - t0
    ms, nl = eng.last_kernel_ms()
    cells = pairs * L * L
    print(f"rep {r}: wall {dt*1e3:.2f} ms kernel {ms:.2f} ms launches {nl}  {cells/ (ms*1e-3)/1e9:.0f} Gcell/s  {pairs/(ms*1e-3)/1e6:.1f} Mpair/s")
print("checksum", int(out.to(torch.int64).sum().item()))
