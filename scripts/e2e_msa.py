#!/usr/bin/env python3
"""BASELINE.json's second metric clause, "end-to-end MSA wall time vs CPU" (GPU box; measurement infrastructure only).

For every set and guide-tree method: the reference's whole CFAMSA::ComputeMSA from oracle/_ref/libfamsa_msa.so (its own
sources; oracle/msa_harness.cpp reads the reference's stage timers) --
  cpu:  as it stands (sort, guide tree with the reference's AVX2 CLCSBP, progressive alignment, refinement);
  gpu:  the same ComputeMSA with `-gt import` of the tree `famsa-gpu -gt <m> -gt_export` wrote for the same input: the tree
        stage is famsa-gpu's, everything downstream the reference's object code on an identical tree.
The two alignments are compared byte for byte.  Writes gpurun_out/e2e_msa_<tag>.json.

    python scripts/e2e_msa.py r06 [threads]
"""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from famsa_amd import seqio  # noqa: E402

TAG = sys.argv[1] if len(sys.argv) > 1 else "rXX"
THREADS = int(sys.argv[2]) if len(sys.argv) > 2 else min(32, os.cpu_count() or 1)
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfamsa_msa.so"))
lib.msa_run.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.c_char_p, C.c_char_p, C.c_int]
CLI = os.path.join(ROOT, "famsa_amd", "famsa-gpu")


def msa(fasta, opts, out):
    t = (C.c_double * 6)()
    err = C.create_string_buffer(1024)
    n = lib.msa_run(fasta.encode(), opts.encode(), t, out.encode(), err, 1024)
    if n < 0:
        raise RuntimeError(err.value.decode())
    return dict(zip(("load", "sort", "tree", "alignment", "refinement", "compute_msa"), [round(x, 4) for x in t]))


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def gpu_tree(fasta, gt, out):
    t0 = time.time()
    p = subprocess.run([CLI, "-v", "-t", str(THREADS), "-gt", gt, "-gt_export", fasta, out], stderr=subprocess.PIPE, text=True, check=True)
    wall = time.time() - t0
    stage = {m.group(1): float(m.group(2)) for m in re.finditer(r"time\.(\w+)=([\d.e+-]+)", p.stderr)}
    return wall, stage


def main():
    sets = [("adeno_fiber (242 seqs)", os.path.join(ROOT, "tests/golden/adeno_fiber/adeno_fiber")),
            ("hemopexin (4188 seqs)", os.path.join(ROOT, "tests/golden/hemopexin/hemopexin"))]
    realmix = "/tmp/realmix.fasta"
    seqio.realmix_fasta(os.path.join(ROOT, "tests/golden"), realmix)
    sets.append(("real sets as one input (13774 seqs)", realmix))
    codes, offsets = seqio.synth_uniform(10000, 400)
    seqio.to_fasta(codes, offsets, "/tmp/synth10k.fasta")
    sets.append(("synthetic 10000 x 400 aa (C3)", "/tmp/synth10k.fasta"))
    seqio.family_fasta(50000, 300, "/tmp/family50k.fasta")
    sets.append(("synthetic family 50000 x 210-300 aa", "/tmp/family50k.fasta"))
    rows = []
    for name, fasta in sets:
        for gt in ("sl", "upgma"):
            cpu = msa(fasta, f"-gt {gt} -t {THREADS}", "/tmp/msa_cpu.afa")
            wall, stage = gpu_tree(fasta, gt, "/tmp/msa_tree.dnd")
            imp = msa(fasta, f"-gt import /tmp/msa_tree.dnd -t {THREADS}", "/tmp/msa_gpu.afa")
            same = sha("/tmp/msa_cpu.afa") == sha("/tmp/msa_gpu.afa")
            cpu_total = cpu["sort"] + cpu["tree"] + cpu["alignment"] + cpu["refinement"]
            # the GPU run's pieces: famsa-gpu's whole command up to the tree (start-up, load, sort, upload, tree), then the
            # reference's alignment and refinement on that tree (the import run's own tree time = parsing the Newick)
            gpu_total = wall + imp["alignment"] + imp["refinement"]
            rows.append({"set": name, "gt": gt, "threads": THREADS, "identical_alignment": same,
                         "cpu": cpu, "cpu_total_s": round(cpu_total, 4), "cpu_tree_share": round(cpu["tree"] / cpu_total, 4),
                         "gpu_tree_command_wall_s": round(wall, 4), "gpu_tree_stage_s": stage.get("tree_build"),
                         "gpu_tree_command_stages": stage,
                         "downstream_on_gpu_tree": {k: imp[k] for k in ("tree", "alignment", "refinement")},
                         "gpu_total_s": round(gpu_total, 4), "speedup_end_to_end": round(cpu_total / gpu_total, 3),
                         "speedup_tree_stage": round(cpu["tree"] / stage["tree_build"], 2) if stage.get("tree_build") else None})
            print(json.dumps(rows[-1]), flush=True)
    out = {"what": "end-to-end MSA: the reference's ComputeMSA stage timers, with its own tree stage (cpu) and on the tree famsa-gpu wrote "
                   "(gpu); alignment + refinement are the reference's object code in both",
           "threads": THREADS, "rows": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"e2e_msa_{TAG}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
