#!/usr/bin/env python3
"""End-to-end wall time of famsa-gpu -gt <m> -gt_export on a synthetic set (dev tool)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from famsa_amd import seqio
n = int(sys.argv[1]); L = int(sys.argv[2]) if len(sys.argv) > 2 else 400
codes, offsets = seqio.synth_uniform(n, L)
f = f"/tmp/synth_{n}_{L}.fasta"
seqio.to_fasta(codes, offsets, f)
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "famsa_amd", "famsa-gpu")
for gt in sys.argv[3:] or ["sl"]:
    t0 = time.time()
    p = subprocess.run([cli, "-v", "-gt", gt, "-gt_export", f, f"/tmp/out_{gt}.dnd"], stderr=subprocess.PIPE, text=True)
    print(gt, "rc", p.returncode, "wall %.2f s" % (time.time() - t0), p.stderr.replace("\n", " "))
