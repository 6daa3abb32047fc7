#!/usr/bin/env python3
"""famsa-gpu on a large synthetic 'family' set (dev tool): -medoidtree -gt upgma etc."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from famsa_amd import seqio
n = int(sys.argv[1]); L = int(sys.argv[2])
f = f"/tmp/family_{n}_{L}.fasta"
if not os.path.exists(f):
    t0 = time.time()
    rng = np.random.Generator(np.random.PCG64(1234))
    anc = rng.integers(0, 20, size=L, dtype=np.uint8)
    A = np.frombuffer(seqio.ALPHABET.encode(), dtype=np.uint8)
    with open(f, "wb") as out:
        B = 20000
        for b0 in range(0, n, B):
            m = min(B, n - b0)
            S = np.tile(anc, (m, 1))
            mut = rng.random((m, L)) < 0.25
            S[mut] = rng.integers(0, 20, size=int(mut.sum()), dtype=np.uint8)
            lens = rng.integers(int(L * 0.7), L + 1, size=m)
            for i in range(m):
                out.write(b">s%d\n" % (b0 + i))
                out.write(A[S[i, : lens[i]]].tobytes())
                out.write(b"\n")
    print("generated", f, "in %.1f s" % (time.time() - t0))
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "famsa_amd", "famsa-gpu")
args = sys.argv[3:] or ["-medoidtree", "-gt", "upgma"]
t0 = time.time()
p = subprocess.run([cli, "-v", *args, "-gt_export", f, "/tmp/family_out.dnd"], stderr=subprocess.PIPE, text=True)
print(" ".join(args), "rc", p.returncode, "wall %.2f s" % (time.time() - t0), p.stderr.replace("\n", " ")[-1500:])
