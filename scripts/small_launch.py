#!/usr/bin/env python3
"""Small sets: the LCS triangle of hemopexin (4188 sequences of 21-210 residues = 7 half-word classes; small neighbouring
classes share a launch since round 5) and of adeno_fiber: kernel time (HIP events), launches and host wall time of the call.
python scripts/small_launch.py"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import famsa_amd
from famsa_amd import seqio
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind
o = oracle_bind.Oracle()
out = {}
eng = famsa_amd.LcsGpu(0)
for name in ("hemopexin/hemopexin", "adeno_fiber/adeno_fiber"):
    seqs = []
    for line in open(os.path.join(ROOT, "tests", "golden", name)):
        if line.startswith(">"): seqs.append("")
        else: seqs[-1] += line.strip()
    enc = [o.encode(s) for s in seqs]
    enc = [enc[i] for i in seqio.sort_order(enc)]
    eng.upload_seqs(enc)
    eng.lcs_triangle()
    ms, wall = [], []
    for _ in range(20):
        t0 = time.perf_counter(); eng.lcs_triangle(); wall.append((time.perf_counter() - t0) * 1e3)
        ms.append(eng.last_kernel_ms()[0])
    out[name.split("/")[0]] = {"n": len(enc), "kernel_ms_median": round(float(np.median(ms)), 3), "call_ms_median": round(float(np.median(wall)), 3),
                              "launches": eng.last_kernel_ms()[1]}
print(json.dumps(out))
