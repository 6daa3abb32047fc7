"""Sequence input helpers shared by tests and bench.py: FASTA reading with the reference's
line handling (reference core/io_service.h:84-127) and the deterministic synthetic sets of
BASELINE.md.  Pure host-side data plumbing; encoding goes through lcsgpu_encode."""
import numpy as np

ALPHABET = "ARNDCQEGHILKMFPSTWYVBZX*"  # reference core/sequence.cpp:17


def read_fasta(path):
    """Returns (ids, residues) as lists of str; ids keep the leading '>'."""
    ids, seqs = [], []
    cur_id, cur = "", []
    with open(path, "rb") as f:
        data = f.read().decode("latin-1")
    for line in data.split("\n"):
        line = line.rstrip("\r\n")
        if not line:
            continue
        if line[0] == ">":
            if cur_id and cur:
                ids.append(cur_id)
                seqs.append("".join(cur))
                cur = []
            cur_id = line
        else:
            cur.append(line)
    if cur_id and cur:
        ids.append(cur_id)
        seqs.append("".join(cur))
    return ids, seqs


def pack(seqs):
    """list of uint8 arrays -> (codes, offsets[n+1] uint64)."""
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    codes = np.concatenate(seqs).astype(np.uint8) if len(seqs) and offsets[-1] > 0 else np.zeros(0, np.uint8)
    return codes, offsets


def sort_order(seqs):
    """Order of CFAMSA::sortAndExtendSequences (reference msa.cpp:245-258): stable sort by
    length descending, then lexicographic over the symbol codes ascending."""
    keys = [(-len(s), bytes(s)) for s in seqs]
    return sorted(range(len(seqs)), key=lambda i: keys[i])


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synth_uniform(n, length=400, seed=None):
    """n sequences of fixed length, residues i.i.d. uniform over the 20 valid codes.
    Counter-based splitmix64 stream, seed 0xFA15A + n."""
    if seed is None:
        seed = 0xFA15A + n
    with np.errstate(over="ignore"):
        idx = np.arange(n * length, dtype=np.uint64)
        z = _splitmix64(idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed))
        z = _splitmix64(z)
    codes = ((z >> np.uint64(11)) % np.uint64(20)).astype(np.uint8)
    offsets = (np.arange(n + 1, dtype=np.uint64) * np.uint64(length)).astype(np.uint64)
    return codes, offsets


def synth_family(n, length=400, seed=None, sub=0.25, indel=0.05):
    """'Family' set: one random ancestor, every member = ancestor with `sub` substitutions and
    `indel` insertions/deletions; returned sorted like the reference sorts (length desc)."""
    if seed is None:
        seed = 0xFA15A + n + 7
    rng = np.random.Generator(np.random.PCG64(seed))
    anc = rng.integers(0, 20, size=length, dtype=np.uint8)
    seqs = []
    for _ in range(n):
        s = anc.copy()
        m = rng.random(length) < sub
        s[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
        keep = rng.random(length) >= indel / 2
        s = s[keep]
        n_ins = int(rng.binomial(len(s), indel / 2))
        pos = np.sort(rng.integers(0, len(s) + 1, size=n_ins))
        s = np.insert(s, pos, rng.integers(0, 20, size=n_ins, dtype=np.uint8))
        seqs.append(s.astype(np.uint8))
    order = sort_order(seqs)
    return [seqs[i] for i in order]


def to_fasta(codes, offsets, path, prefix="s"):
    with open(path, "w") as f:
        for i in range(len(offsets) - 1):
            s = codes[int(offsets[i]):int(offsets[i + 1])]
            f.write(f">{prefix}{i}\n")
            f.write("".join(ALPHABET[c] for c in s))
            f.write("\n")


def family_fasta(n, length, path, seed=1234):
    """Large 'family' FASTA written in blocks (the C5 shape: one ancestor, 25 % substitutions, member
    lengths uniform in [0.7*length, length]), ids s<i>.  Deterministic; used by the at-size tests, the
    golden generator (oracle/make_golden_large.py) and the end-to-end timing scripts."""
    rng = np.random.Generator(np.random.PCG64(seed))
    anc = rng.integers(0, 20, size=length, dtype=np.uint8)
    A = np.frombuffer(ALPHABET.encode(), dtype=np.uint8)
    with open(path, "wb") as out:
        B = 20000
        for b0 in range(0, n, B):
            m = min(B, n - b0)
            S = np.tile(anc, (m, 1))
            mut = rng.random((m, length)) < 0.25
            S[mut] = rng.integers(0, 20, size=int(mut.sum()), dtype=np.uint8)
            lens = rng.integers(int(length * 0.7), length + 1, size=m)
            rows = A[S]  # residue letters, one row per member
            out.write(b"".join(b">s%d\n%s\n" % (b0 + i, rows[i, : lens[i]].tobytes()) for i in range(m)))
    return path


def family_set(n, length, seed=1234):
    """The sequences family_fasta writes (same generator, same seed: one ancestor, 25 % substitutions, member lengths
    uniform in [0.7*length, length]), in memory: (codes, offsets), input order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    anc = rng.integers(0, 20, size=length, dtype=np.uint8)
    parts, lens_all = [], []
    B = 20000
    for b0 in range(0, n, B):
        m = min(B, n - b0)
        S = np.tile(anc, (m, 1))
        mut = rng.random((m, length)) < 0.25
        S[mut] = rng.integers(0, 20, size=int(mut.sum()), dtype=np.uint8)
        lens = rng.integers(int(length * 0.7), length + 1, size=m)
        keep = np.arange(length)[None, :] < lens[:, None]
        parts.append(S[keep])
        lens_all.append(lens)
    lens = np.concatenate(lens_all).astype(np.uint64)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    return np.concatenate(parts).astype(np.uint8), offsets


def reorder(codes, offsets, order):
    """The set with sequence k = the old sequence order[k]."""
    o = offsets.astype(np.int64)
    lens = (o[1:] - o[:-1])[order]
    new_off = np.zeros(len(order) + 1, dtype=np.uint64)
    np.cumsum(lens.astype(np.uint64), out=new_off[1:])
    idx = np.concatenate([np.arange(o[i], o[i + 1]) for i in order]) if len(order) else np.zeros(0, np.int64)
    return codes[idx], new_off


REALMIX_PARTS = [("af", "adeno_fiber/adeno_fiber"), ("hp", "hemopexin/hemopexin"), ("afd", "adeno_fiber_duplicates/adeno_fiber_duplicates"),
                 ("hpd", "hemopexin_duplicates/hemopexin_duplicates"), ("afx", "adeno_fiber_extra/adeno_fiber_extra")]


def realmix_fasta(golden_dir, path):
    """One FASTA from the upstream REAL sets held as fixtures under tests/golden (the reference's test/ directory):
    adeno_fiber + hemopexin + both duplicates sets + adeno_fiber_extra (non-standard residue symbols J, U) = 13 774
    records, 21-210 residues, exact duplicates (x2 / x3) and heavy distance ties included -- the short, ragged,
    tie-heavy regime no synthetic set has.  Ids get a per-set prefix so that every record name is unique; the record
    order is the files' order, the sets one after the other.  Returns the number of records."""
    n = 0
    with open(path, "w") as out:
        for tag, rel in REALMIX_PARTS:
            ids, seqs = read_fasta(golden_dir + "/" + rel)
            for i, (name, res) in enumerate(zip(ids, seqs)):
                out.write(f">{tag}{i}|{name[1:]}\n{res}\n")
                n += 1
    return n
