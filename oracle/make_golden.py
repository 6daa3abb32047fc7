#!/usr/bin/env python3
"""Generate tests/golden/ fixtures from the REFERENCE itself (oracle/_ref/libfamsa_ref.so, built
by oracle/Makefile from /root/reference).  Run in the build container only; the outputs are data
(inputs + expected outputs) and are committed.  Usage: python oracle/make_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind  # noqa: E402
from famsa_amd import seqio  # noqa: E402

G = oracle_bind.GOLDEN


def adversarial_set():
    """Inputs the reference's own tests do not hold: carry-quirk homopolymers,
    word-boundary lengths, the unrolled<->loop switch at 32 words, non-matching symbols, lower case,
    foreign characters, an all-X sequence, exact duplicates."""
    rng = np.random.Generator(np.random.PCG64(20250801))
    A = seqio.ALPHABET

    def rnd(n, k=20):
        return "".join(A[i] for i in rng.integers(0, k, size=n))

    seqs = []
    for n in (65, 128, 129, 192, 256, 448):  # homopolymer refs (note Q)
        seqs.append("A" * n)
    seqs += ["A", "AA", "A" * 9, "C" * 64, "C" * 63, "W" * 200 + rnd(50)]
    seqs.append(rnd(64) + "K" * 64 + rnd(30))          # homopolymer word at index 1 inside a real sequence
    seqs.append(rnd(60) + "K" * 70 + rnd(30))          # not aligned -> no quirk
    seqs.append(rnd(128) + "L" * 128 + rnd(17))        # two quirk words
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 511, 512, 513, 1023, 1025, 2047, 2048):
        seqs.append(rnd(n))
    seqs.append(rnd(300, 24))                            # B Z X * mixed in
    seqs.append("X" * 50)                                # nothing matches -> lcs 0
    seqs.append("BZX*" * 20)
    seqs.append(rnd(120).lower())                        # lower case folds to upper
    seqs.append("ARND?JOU" + rnd(40) + "acdefghiklmnpqrstvwy")  # foreign symbols -> 22
    s = rnd(150)
    seqs += [s, s, s[:100], s[50:]]                      # duplicates / substrings
    seqs += [rnd(400) for _ in range(8)]
    ids = [f">adv{i}" for i in range(len(seqs))]
    return ids, seqs


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = oracle_bind.Ref()
    meta = {}

    # 1. adversarial set: full oriented square matrix, classic and AVX2 dispatch
    ids, seqs = adversarial_set()
    with open(os.path.join(G, "adversarial.fasta"), "w") as f:
        for i, s in zip(ids, seqs):
            f.write(f"{i}\n{s}\n")
    h = ref.open_seqs(ids, seqs)
    n = len(seqs)
    allids = np.arange(n)
    m0 = ref.lcs_rect(h, allids, allids, isa=0)
    m2 = ref.lcs_rect(h, allids, allids, isa=2)
    # the reference's AVX2 lanes and classic path agree except that the full groups of 8 go
    # through AVX2; both are stored
    np.savez_compressed(os.path.join(G, "adversarial_lcs.npz"), classic=m0.astype(np.uint16),
                        avx2=m2.astype(np.uint16))
    meta["adversarial"] = {"n": n, "classic_eq_avx2": bool((m0 == m2).all())}
    ref.dist_export(h, os.path.join(G, "adversarial_dist_sq.csv"), square=True)
    ref.dist_export(h, os.path.join(G, "adversarial_pid.csv"), pid=True)
    ref.close(h)
    # trees: the reference itself breaks (crash in UPGMA, degenerate NJ) once a pair has LCS 0
    # (distance ~FLT_MAX), so tree goldens use the members with a rich residue set only
    keep = [i for i, s_ in enumerate(seqs) if len(set(s_.upper()) & set(seqio.ALPHABET[:20])) >= 10]
    tids, tseqs = [ids[i] for i in keep], [seqs[i] for i in keep]
    with open(os.path.join(G, "adversarial_tree.fasta"), "w") as f:
        for i, s_ in zip(tids, tseqs):
            f.write(f"{i}\n{s_}\n")
    h = ref.open_seqs(tids, tseqs)
    for gt in ("sl", "slink", "upgma", "nj"):
        open(os.path.join(G, f"adversarial_tree_{gt}.dnd"), "wb").write(ref.tree(h, gt))
    ref.close(h)

    # 1b. long sequences (> 2048 residues: the reference's LoopCalculate path), with and without
    #     carry-quirk words beyond the first 32-word segment
    rng = np.random.Generator(np.random.PCG64(77))
    A = seqio.ALPHABET

    def rnd(n_):
        return "".join(A[i] for i in rng.integers(0, 20, size=n_))

    lseqs = [rnd(2049), rnd(2112), rnd(3000), rnd(4096), rnd(4097), rnd(6500),
             rnd(2048) + "M" * 64 + rnd(100),            # quirk word exactly at word 32 (segment edge)
             rnd(2048 + 640) + "G" * 128 + rnd(777),     # quirk words inside segment 1
             rnd(100), rnd(2048), "M" * 300, rnd(5000)[:4500] + "G" * 70]
    lids = [f">long{i}" for i in range(len(lseqs))]
    with open(os.path.join(G, "adversarial_long.fasta"), "w") as f:
        for i, s_ in zip(lids, lseqs):
            f.write(f"{i}\n{s_}\n")
    h = ref.open_seqs(lids, lseqs)
    ln = len(lseqs)
    lm0 = ref.lcs_rect(h, np.arange(ln), np.arange(ln), isa=0)
    lm2 = ref.lcs_rect(h, np.arange(ln), np.arange(ln), isa=2)
    np.savez_compressed(os.path.join(G, "adversarial_long_lcs.npz"), classic=lm0.astype(np.uint16),
                        avx2=lm2.astype(np.uint16))
    meta["adversarial_long"] = {"n": ln, "classic_eq_avx2": bool((lm0 == lm2).all())}
    ref.close(h)

    # 2. adeno_fiber: full oriented square LCS (input order)
    h = ref.open_fasta(os.path.join(G, "adeno_fiber", "adeno_fiber"))
    n = ref.lib.ref_count(h)
    allids = np.arange(n)
    m = ref.lcs_rect(h, allids, allids, isa=2)
    np.savez_compressed(os.path.join(G, "adeno_fiber", "lcs_square.npz"), lcs=m.astype(np.uint16))
    open(os.path.join(G, "adeno_fiber", "nj.dnd"), "wb").write(ref.tree(h, "nj"))
    ref.close(h)

    # 3. hemopexin: plain (non-medoid) trees + checksums of the full triangle / csv + sample rows
    h = ref.open_fasta(os.path.join(G, "hemopexin", "hemopexin"))
    n = ref.lib.ref_count(h)
    for gt in ("sl", "slink", "upgma", "nj"):
        open(os.path.join(G, "hemopexin", f"{gt}.dnd"), "wb").write(ref.tree(h, gt, threads=8))
    rows = np.array(sorted(set(list(range(0, n, 97)) + [1, 2, 63, 64, 65, n - 1])))
    m = ref.lcs_rect(h, rows, np.arange(n), isa=2)
    np.savez_compressed(os.path.join(G, "hemopexin", "lcs_rows.npz"), rows=rows, lcs=m.astype(np.uint16))
    tmp = "/tmp/hemopexin_dist.csv"
    ref.dist_export(h, tmp, threads=8)
    meta["hemopexin"] = {"n": n, "dist_csv_sha256": hashlib.sha256(open(tmp, "rb").read()).hexdigest(),
                         "dist_csv_bytes": os.path.getsize(tmp)}
    # full input-order triangle through the reference (ref = row i, partner = col j < i)
    tri = np.empty(n * (n - 1) // 2, np.uint16)
    for i in range(1, n):
        tri[i * (i - 1) // 2: i * (i - 1) // 2 + i] = ref.lcs_rect(h, [i], np.arange(i), isa=2)[0]
    meta["hemopexin"]["triangle_u16_sha256"] = sha(tri)
    ref.close(h)

    # 4. synthetic slice (2000 x 400 uniform, the bench generator) : trees + triangle checksum
    codes, offsets = seqio.synth_uniform(2000, 400)
    seqio.to_fasta(codes, offsets, "/tmp/synth2k.fasta")
    h = ref.open_fasta("/tmp/synth2k.fasta")
    for gt in ("sl", "upgma"):
        open(os.path.join(G, f"synth2k_{gt}.dnd"), "wb").write(ref.tree(h, gt, threads=8))
    tri = np.empty(2000 * 1999 // 2, np.uint16)
    for i in range(1, 2000):
        tri[i * (i - 1) // 2: i * (i - 1) // 2 + i] = ref.lcs_rect(h, [i], np.arange(i), isa=2)[0]
    meta["synth2k"] = {"triangle_u16_sha256": sha(tri), "codes_sha256": sha(codes)}
    ref.close(h)

    json.dump(meta, open(os.path.join(G, "meta.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
