#!/usr/bin/env python3
"""GPU box: lcsgpu_upgma beyond 100 000 sequences -- the batched form on n x (n + spare) slots against one launch per merge on
the n x n matrix (two forms that share the distance kernels and nothing of the merge phase), the device memory both hold,
their times.  usage: upgma_beyond.py [n=250000] [len=400] -> gpurun_out/upgma_beyond.txt"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import famsa_amd
from famsa_amd import seqio

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 400
out = open(os.path.join(ROOT, "gpurun_out", "upgma_beyond.txt"), "w")


def say(*a):
    print(*a, flush=True)
    print(*a, file=out, flush=True)


def vram():
    try:
        t = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True, timeout=20).stdout
        return [int(ln.split(":")[-1]) for ln in t.splitlines() if "Used" in ln][0] / 1e9
    except Exception:
        return float("nan")


codes, offsets = seqio.synth_uniform(n, L)
res = {}
for name, env in (("batches of 32 merges, n x (n + spare) slots, compaction", {"LCSGPU_UPGMA_BATCH": "32"}),
                  ("one launch per merge, n x n matrix", {"LCSGPU_UPGMA_BATCH": "0"})):
    os.environ.update(env)
    os.environ["LCSGPU_PROFILE"] = "1"
    eng = famsa_amd.LcsGpu(0)
    eng.upload(codes, offsets)
    t0 = time.time()
    try:
        left, right = eng.upgma(1, False)
        res[name] = hashlib.sha256(left.tobytes() + right.tobytes()).hexdigest()
        say(f"n = {n} x {L} aa, {name}: {time.time() - t0:.2f} s, device memory in use afterwards {vram():.1f} GB, tree sha256 {res[name][:16]}")
    except famsa_amd.LcsGpuError as e:
        say(f"n = {n} x {L} aa, {name}: {e}")
    eng.close()
    del eng
    time.sleep(8)
vals = list(res.values())
say("the two forms agree" if len(vals) == 2 and vals[0] == vals[1] else f"NO AGREEMENT / one form did not run: {res}")
