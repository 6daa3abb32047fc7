#!/usr/bin/env python3
"""Dev tool: duration of lcsgpu_row_minima_dev over the whole triangle of the bench set (n x 400 aa)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, famsa_amd
from famsa_amd import seqio
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
codes, offsets = seqio.synth_uniform(n, 400)
eng = famsa_amd.LcsGpu(0); eng.upload(codes, offsets)
pairs = n * (n - 1) // 2
tri = torch.empty(pairs, dtype=torch.int16, device="cuda:0")
out = torch.zeros(2 * n, dtype=torch.float64, device="cuda:0")
eng.lcs_triangle_dev(0, n, tri.data_ptr(), 2, sync=True)
for kind in (1, 0):
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.row_minima_dev(tri.data_ptr(), 2, 0, n, kind, out.data_ptr(), sync=True)
        dt = time.perf_counter() - t0
    print(f"kind {kind}: {dt*1e3:.2f} ms  {2*pairs/dt/1e12:.2f} TB/s  checksum {float(out[::2][1:].sum()):.6f} {int(out[1::2].view(torch.int64).sum())}")
