#!/usr/bin/env python3
"""Small sets: the LCS triangle of hemopexin (4188 sequences of 21-210 residues = 7 half-word classes = 7 launches) and of
adeno_fiber, kernel time (HIP events) and host wall time of the call, one launch after the other (default) and with the
launches spread over side streams (LCSGPU_SPREAD=1).  python scripts/small_launch.py"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for env in ({}, {"LCSGPU_SPREAD": "1"}):
        e = dict(os.environ); e.update(env)
        print(subprocess.run([sys.executable, __file__, "run"], env=e, stdout=subprocess.PIPE, text=True).stdout.strip())
    sys.exit(0)
import numpy as np
import famsa_amd
from famsa_amd import seqio
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind
o = oracle_bind.Oracle()
out = {"spread": "LCSGPU_SPREAD" in os.environ}
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
