#!/usr/bin/env python3
"""Back-to-back launches without host sync in between (separates clock ramp / launch gaps from kernel time)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import famsa_amd
from famsa_amd import seqio
n = int(sys.argv[1]); L = int(sys.argv[2]); reps = int(sys.argv[3])
codes, offsets = seqio.synth_uniform(n, L)
eng = famsa_amd.LcsGpu(0); eng.upload(codes, offsets)
pairs = n * (n - 1) // 2
out = torch.empty(pairs, dtype=torch.int16, device="cuda:0")
eng.lcs_triangle_dev(0, n, out.data_ptr(), 2, sync=True)
for trial in range(3):
    t0 = time.time()
    for r in range(reps):
        eng.lcs_triangle_dev(0, n, out.data_ptr(), 2, sync=False)
    eng.sync()
    dt = (time.time() - t0) / reps
    print(f"n={n} x{reps} back-to-back: {dt*1e3:.2f} ms per launch  {pairs/dt/1e6:.1f} Mpair/s  {pairs*L*L/dt/1e9:.0f} Gcell/s")
