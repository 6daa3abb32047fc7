#!/usr/bin/env python3
"""dev tool (GPU box): LCS kernel time of each row block of an N-way split, one after the other on one GPU --
what the slowest rank of a real N-GPU run would take.  usage: block_balance.py [n=100000] [len=400]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import famsa_amd
from famsa_amd import seqio
from famsa_amd.rowblock import row_cuts, pairs_in_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 400
codes, offsets = seqio.synth_uniform(n, L)
eng = famsa_amd.LcsGpu(0)
eng.upload(codes, offsets)
tri = torch.empty(n * (n - 1) // 2, dtype=torch.int16, device="cuda:0")
torch.cuda.synchronize()
for world in (1, 2, 4, 8):
    cuts = row_cuts(n, world)
    ms = []
    for rep in range(2):
        ms = []
        for r in range(world):
            eng.lcs_triangle_dev(cuts[r], cuts[r + 1], tri.data_ptr(), 2)
            ms.append(eng.last_kernel_ms()[0])
    total = sum(ms)
    print("N=%d  blocks(ms): %s  max/mean=%.3f  sum=%.1f  ideal-speedup %.2f of %d" %
          (world, " ".join("%.1f" % m for m in ms), max(ms) / (total / world), total, total / max(ms), world), flush=True)
