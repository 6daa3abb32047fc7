#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer form (lcsgpu_lcs_triangle: LCS + D2H of 2 B/pair) -- measurement tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import famsa_amd
from famsa_amd import seqio
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
codes, offsets = seqio.synth_uniform(n, 400)
eng = famsa_amd.LcsGpu(0)
eng.upload(codes, offsets)
pairs = n * (n - 1) // 2
for rep in range(2):
    t0 = time.time()
    tri = eng.lcs_triangle(0, n)
    dt = time.time() - t0
    ms, _ = eng.last_kernel_ms()
    print(f"rep {rep}: n={n} host-buffer call {dt:.3f} s (kernel {ms/1e3:.3f} s) -> {pairs/dt/1e9:.2f} Gpair/s PCIe-inclusive, "
          f"{pairs*160000/dt/1e12:.0f} Tcell/s; D2H {pairs*2/1e9:.1f} GB", flush=True)
    del tri
