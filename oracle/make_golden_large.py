#!/usr/bin/env python3
"""At-size fixtures (BASELINE configs C3, C4, C5) from the REFERENCE itself
(oracle/_ref/libfamsa_ref.so = /root/reference sources built by oracle/Makefile).  Build container
only; minutes of CPU.  Writes tests/golden/meta_large.json (sha256 values only -- the data is
regenerated deterministically by famsa_amd/seqio.py on the GPU box).

    python oracle/make_golden_large.py [c3] [c4] [c5] [c5huge] [c4upgma] [realmix] [c4slink] [c3indel] [c4indel] [realmixnj]
                                       [familyindel] [realmixindel]
                                                                                    (default: c3 c5 c4)

c5huge = the 3 000 000-sequence family set (BASELINE config C5's size); c4upgma = -gt upgma / upgma_modified at
100 000 x 400 aa (the reference holds the 20 GB float triangle in host memory).

Model: the reference's own at-size regression, .github/workflows/self-hosted.yml:424-461."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind  # noqa: E402
from famsa_amd import seqio  # noqa: E402

OUT = os.environ.get("GOLDEN_OUT") or os.path.join(oracle_bind.GOLDEN, "meta_large.json")  # (GOLDEN_OUT: a second generator running next to the first)
THREADS = len(os.sched_getaffinity(0))


def sha(b):
    return hashlib.sha256(b).hexdigest()


def load():
    return json.load(open(OUT)) if os.path.exists(OUT) else {}


def save(meta):
    """Merge into what the file holds now (another generator may have added sections or keys meanwhile), then write."""
    cur = load()
    for section, rec in meta.items():
        if isinstance(rec, dict) and isinstance(cur.get(section), dict):
            cur[section].update(rec)
        else:
            cur[section] = rec
    json.dump(cur, open(OUT, "w"), indent=1, sort_keys=True)


def c3(ref, meta):
    n, L = 10000, 400
    codes, offsets = seqio.synth_uniform(n, L)
    path = "/tmp/golden_synth10k.fasta"
    seqio.to_fasta(codes, offsets, path)
    h = ref.open_fasta(path)
    rec = {"n": n, "len": L, "codes_sha256": sha(codes.tobytes())}
    t0 = time.time()
    hh = hashlib.sha256()  # rows in order = the packed triangle, ref = row i, partner = column j < i
    step = 500  # ref_lcs_rect copies the whole set per call: ask for many rows at a time
    for i0 in range(0, n, step):
        i1 = min(n, i0 + step)
        m = ref.lcs_rect(h, np.arange(i0, i1), np.arange(i1 - 1), isa=2).astype(np.uint16)
        for i in range(max(i0, 1), i1):
            hh.update(m[i - i0, :i].tobytes())
    rec["triangle_u16_sha256"] = hh.hexdigest()
    print("c3 triangle %.0f s" % (time.time() - t0), flush=True)
    for gt in ("sl", "slink", "upgma"):
        t0 = time.time()
        rec[f"{gt}_newick_sha256"] = sha(ref.tree(h, gt, threads=THREADS))
        print("c3", gt, "%.0f s" % (time.time() - t0), flush=True)
    ref.close(h)
    meta["synth10k"] = rec
    save(meta)


def c3more(ref, meta):
    """-gt nj (O(n^3) on one thread in the reference: minutes) and upgma_modified at 10 000 x 400 aa."""
    n, L = 10000, 400
    codes, offsets = seqio.synth_uniform(n, L)
    path = "/tmp/golden_synth10k.fasta"
    seqio.to_fasta(codes, offsets, path)
    h = ref.open_fasta(path)
    rec = meta["synth10k"]
    for gt in ("upgma_modified", "nj"):
        t0 = time.time()
        rec[f"{gt}_newick_sha256"] = sha(ref.tree(h, gt, threads=THREADS))
        print("c3", gt, "%.0f s" % (time.time() - t0), flush=True)
        save(meta)
    ref.close(h)


def c4(ref, meta, gts=("sl",), distance=1, n=100000, key="synth100k"):
    """-gt <gts> at n x 400 aa; distance 0 = -dist indel_div_lcs (keys carry "_indel")."""
    L = 400
    codes, offsets = seqio.synth_uniform(n, L)
    path = f"/tmp/golden_synth{n}.fasta"
    seqio.to_fasta(codes, offsets, path)
    h = ref.open_fasta(path)
    rec = meta.get(key, {"n": n, "len": L, "codes_sha256": sha(codes.tobytes())})
    tag = "" if distance == 1 else "_indel"
    for gt in gts:
        t0 = time.time()
        rec[f"{gt}{tag}_newick_sha256"] = sha(ref.tree(h, gt, distance=distance, threads=THREADS))
        rec[f"{gt}{tag}_reference_seconds_{THREADS}_threads"] = round(time.time() - t0, 1)
        print(key, gt + tag, "%.0f s" % (time.time() - t0), flush=True)
        meta[key] = rec
        save(meta)
    ref.close(h)


def c5(ref, meta, sizes=(200000, 1000000)):
    for n in sizes:
        path = f"/tmp/golden_family_{n}_300.fasta"
        seqio.family_fasta(n, 300, path)
        fh = hashlib.sha256()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                fh.update(blk)
        h = ref.open_fasta(path)
        t0 = time.time()
        nw = ref.tree(h, "upgma", heuristic=2, threads=THREADS, cap=max(1 << 25, 24 * n))  # -medoidtree -gt upgma, default parameters
        rec = {"n": n, "len": 300, "fasta_sha256": fh.hexdigest(), "medoid_upgma_newick_sha256": sha(nw),
               "newick_bytes": len(nw), f"reference_seconds_{THREADS}_threads": round(time.time() - t0, 1)}
        print("c5", n, "%.0f s" % (time.time() - t0), flush=True)
        ref.close(h)
        meta[f"family{n}"] = rec
        save(meta)


def realmix(ref, meta):
    """The upstream real sets in one FASTA (famsa_amd/seqio.py: realmix_fasta; 13 774 records): every tree method with
    and without duplicate removal, the medoid heuristic, the distance export -- the reference's own runs."""
    path = "/tmp/golden_realmix.fasta"
    n = seqio.realmix_fasta(oracle_bind.GOLDEN, path)
    rec = {"n": n, "fasta_sha256": sha(open(path, "rb").read())}
    h = ref.open_fasta(path)
    for gt in ("sl", "slink", "upgma", "upgma_modified", "nj"):
        for keep in (0, 1):
            if gt == "nj" and keep:
                continue  # (O(n^3) on one thread in the reference: hours at 13 774)
            t0 = time.time()
            rec[f"{gt}{'_keepdups' if keep else ''}_newick_sha256"] = sha(ref.tree(h, gt, keep_dups=keep, threads=THREADS))
            print("realmix", gt, keep, "%.0f s" % (time.time() - t0), flush=True)
    for gt in ("sl", "upgma"):
        t0 = time.time()
        rec[f"medoid_{gt}_newick_sha256"] = sha(ref.tree(h, gt, heuristic=2, threads=THREADS))
        rec[f"medoid_{gt}_keepdups_newick_sha256"] = sha(ref.tree(h, gt, heuristic=2, keep_dups=1, threads=THREADS))
        print("realmix medoid", gt, "%.0f s" % (time.time() - t0), flush=True)
    out = "/tmp/golden_realmix_dist.csv"
    t0 = time.time()
    ref.dist_export(h, out, threads=THREADS)
    hh = hashlib.sha256()
    with open(out, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            hh.update(blk)
    rec["dist_export_sha256"] = hh.hexdigest()
    rec["dist_export_bytes"] = os.path.getsize(out)
    os.unlink(out)
    print("realmix dist_export %.0f s" % (time.time() - t0), flush=True)
    ref.close(h)
    meta["realmix"] = rec
    save(meta)


def family_indel(ref, meta, n=100000):
    """-dist indel_div_lcs on a RAGGED set (family, lengths 210-300): at one fixed length both distances are monotone in the
    LCS length and single linkage gives the same tree for either (synth10k / synth100k: sl_indel == sl by construction);
    here the two Transforms order the pairs differently (reference tree/AbstractTreeGenerator.hpp:65-75)."""
    path = f"/tmp/golden_family_{n}_300.fasta"
    seqio.family_fasta(n, 300, path)
    h = ref.open_fasta(path)
    key = f"family{n}"
    rec = meta.get(key, {"n": n, "len": 300})
    for gt, distance in (("sl", 0), ("upgma", 0), ("sl", 1)):
        tag = "" if distance == 1 else "_indel"
        t0 = time.time()
        rec[f"{gt}{tag}_newick_sha256"] = sha(ref.tree(h, gt, distance=distance, threads=THREADS, cap=max(1 << 25, 24 * n)))
        rec[f"{gt}{tag}_reference_seconds_{THREADS}_threads"] = round(time.time() - t0, 1)
        print(key, gt + tag, "%.0f s" % (time.time() - t0), flush=True)
        meta[key] = rec
        save(meta)
    ref.close(h)


def realmix_indel(ref, meta):
    """-dist indel_div_lcs -gt sl / upgma on the 13 774-record real set (lengths 21-210)."""
    path = "/tmp/golden_realmix.fasta"
    seqio.realmix_fasta(oracle_bind.GOLDEN, path)
    h = ref.open_fasta(path)
    for gt in ("sl", "upgma"):
        t0 = time.time()
        meta["realmix"][f"{gt}_indel_newick_sha256"] = sha(ref.tree(h, gt, distance=0, threads=THREADS))
        print("realmix", gt, "indel %.0f s" % (time.time() - t0), flush=True)
    ref.close(h)
    save(meta)


def realmix_nj_keepdups(ref, meta):
    """-gt nj -keep-duplicates on the 13 774-record real set (the reference's O(n^3) loop on one thread)."""
    path = "/tmp/golden_realmix.fasta"
    seqio.realmix_fasta(oracle_bind.GOLDEN, path)
    h = ref.open_fasta(path)
    t0 = time.time()
    meta["realmix"]["nj_keepdups_newick_sha256"] = sha(ref.tree(h, "nj", keep_dups=1, threads=THREADS))
    meta["realmix"]["nj_keepdups_reference_seconds"] = round(time.time() - t0, 1)
    print("realmix nj keepdups %.0f s" % (time.time() - t0), flush=True)
    ref.close(h)
    save(meta)


def main():
    which = sys.argv[1:] or ["c3", "c5", "c4"]
    ref = oracle_bind.Ref()
    meta = load()
    for w in which:
        {"c3": c3, "c3more": c3more, "c4": c4, "c5": c5, "c5huge": lambda r, m: c5(r, m, (3000000,)),
         "c4upgma": lambda r, m: c4(r, m, ("upgma", "upgma_modified")), "realmix": realmix,
         "c4slink": lambda r, m: c4(r, m, ("slink",)),
         "c3indel": lambda r, m: c4(r, m, ("sl", "upgma"), distance=0, n=10000, key="synth10k"),
         "c4indel": lambda r, m: c4(r, m, ("sl", "upgma"), distance=0),
         "realmixnj": realmix_nj_keepdups, "familyindel": family_indel, "realmixindel": realmix_indel}[w](ref, meta)
    print(json.dumps(load(), indent=1))


if __name__ == "__main__":
    main()
