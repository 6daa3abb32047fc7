#!/usr/bin/env python3
"""Time lcsgpu_clarans on the MedoidTree default shape (2000-member sample, 100 medoids) -- dev tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import famsa_amd

rng = np.random.default_rng(7)
L = 255
anc = rng.integers(0, 20, size=L, dtype=np.uint8)
seqs = []
for _ in range(2300):
    s = anc.copy()
    m = rng.random(L) < 0.25
    s[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
    seqs.append(s[: int(rng.integers(int(L * 0.7), L + 1))].copy())
eng = famsa_amd.LcsGpu(0)
eng.upload_seqs(seqs)
ids = np.sort(rng.permutation(2300)[:2000]).astype(np.int32)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for r in range(reps):
    t0 = time.time()
    med = eng.clarans(ids, 100)
    print("clarans 2000/100: %.1f ms  medoids[:5]=%s" % (1e3 * (time.time() - t0), med[:5].tolist()))
