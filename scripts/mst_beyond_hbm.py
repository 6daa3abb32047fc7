#!/usr/bin/env python3
"""Single linkage beyond the HBM-resident triangle: lcsgpu_mst_prim on n synthetic sequences, in the mode the library
picks (all rows resident / part of them / none) or a forced one.  Prints one JSON line.
    python scripts/mst_beyond_hbm.py --n 600000 [--len 400] [--mode auto|passes|fused|recompute] [--fake-hbm-gb X]"""
import argparse, hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=600000)
ap.add_argument("--len", type=int, default=400)
ap.add_argument("--mode", default="auto")
ap.add_argument("--fake-hbm-gb", type=float, default=0)
a = ap.parse_args()
os.environ["LCSGPU_PROFILE"] = "1"
if a.mode != "auto":
    os.environ["LCSGPU_MST_MODE"] = a.mode
if a.fake_hbm_gb:
    os.environ["LCSGPU_FAKE_HBM_GB"] = str(a.fake_hbm_gb)
import numpy as np
import famsa_amd
from famsa_amd import seqio
codes, offsets = seqio.synth_uniform(a.n, a.len)
eng = famsa_amd.LcsGpu(0)
t0 = time.perf_counter()
eng.upload(codes, offsets)
t1 = time.perf_counter()
edges = eng.mst_prim(1)
t2 = time.perf_counter()
pairs = a.n * (a.n - 1) // 2
print(json.dumps({"n": a.n, "len": a.len, "mode": a.mode, "upload_s": round(t1 - t0, 3), "mst_s": round(t2 - t1, 3),
                  "pairs": pairs, "tcell_per_s_one_pass_equivalent": round(pairs * a.len * a.len / (t2 - t1) / 1e12, 1),
                  "edges_sha256": hashlib.sha256(np.ascontiguousarray(edges).tobytes()).hexdigest(),
                  "triangle_gb_if_resident": round(pairs * 2 / 1e9, 1)}), flush=True)
eng.close()
