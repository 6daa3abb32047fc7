#!/usr/bin/env python3
"""Dev tool: rate of the triangle on a variable-length set (lengths uniform in [0.7 L, L], random residues),
uploaded in FAMSA's working order (length descending) and in random order; cells = sum of len_i * len_j."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, famsa_amd
from famsa_amd import seqio
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.Generator(np.random.PCG64(9))
lens = rng.integers(int(0.7 * L), L + 1, size=n)
for name, order in (("sorted", np.argsort(-lens, kind="stable")), ("random", np.arange(n))):
    ls = lens[order].astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(ls)]).astype(np.uint64)
    codes = rng.integers(0, 20, size=int(offsets[-1]), dtype=np.uint8)
    eng = famsa_amd.LcsGpu(0); eng.upload(codes, offsets)
    pairs = n * (n - 1) // 2
    out = torch.empty(pairs, dtype=torch.int16, device="cuda:0")
    tot = float(ls.sum()); cells = (tot * tot - float((ls * ls).sum())) / 2
    for r in range(2):
        eng.lcs_triangle_dev(0, n, out.data_ptr(), 2, sync=True)
        ms, nl = eng.last_kernel_ms()
    print(f"{name}: n={n} lengths {ls.min()}..{ls.max()}: {ms:.1f} ms in {nl} launches, {cells/(ms*1e-3)/1e12:.0f} Tcell/s")
    eng.close()
