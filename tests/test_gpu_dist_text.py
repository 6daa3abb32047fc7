"""-dist_export rows as text made on the device (lcsgpu_dist_text_*, famsa_amd/csrc/text_kernels.hip) against the
reference's golden CSV files and against the oracle's LCS + Transform + number format (oracle_format_dist =
NumericConversions::Double2PChar, reference utils/conversion.h:109-119; writer: tree/DistanceCalculator.cpp:88-113).
The bar: byte-identical."""
import os

import numpy as np
import pytest

import oracle_bind
from famsa_amd import lcsgpu, seqio

pytestmark = pytest.mark.gpu
G = oracle_bind.GOLDEN


def load_set(path):
    ids, seqs = seqio.read_fasta(path)
    return ids, [lcsgpu.encode(s) for s in seqs]


def expected_rows(oracle, names, enc, square, pid, kind=1):
    """One bytes object per row, from oracle values only."""
    codes, offsets = seqio.pack(enc)
    n = len(enc)
    lens = [len(e) for e in enc]
    lcs = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    f64 = oracle.lib.oracle_dist_indel075_f64 if kind == 1 else oracle.lib.oracle_dist_indel_f64
    rows = []
    for i in range(n):
        row = [names[i]]
        for j in range(n if square else i):
            if pid:
                v = float(oracle.lib.oracle_pid_f32(int(lcs[i, j]), lens[i], lens[j]))
            else:
                with np.errstate(over="ignore"):
                    v = float(np.float32(f64(int(lcs[i, j]), lens[i], lens[j])))
            row.append(oracle.format_dist(v))
        rows.append((",".join(row) + "\n").encode("latin-1"))
    return rows


def splits(n, kind):
    if kind == "one":
        return [(0, n)]
    if kind == "rows":
        return [(i, i + 1) for i in range(n)]
    rng = np.random.Generator(np.random.PCG64(n))
    cuts = sorted(set([0, n] + list(rng.integers(1, n, size=7))))
    return list(zip(cuts[:-1], cuts[1:]))


@pytest.mark.parametrize("name,square,pid", [("dist", False, False), ("pid", False, True),
                                             ("dist_sq", True, False), ("pid_sq", True, True)])
@pytest.mark.parametrize("split", ["one", "ragged"])
def test_adeno_goldens(engine, name, square, pid, split):
    """The reference's own test/adeno_fiber/*.csv, block by block."""
    ids, enc = load_set(os.path.join(G, "adeno_fiber", "adeno_fiber"))
    engine.upload_seqs(enc)
    gold = open(os.path.join(G, "adeno_fiber", name + ".csv"), "rb").read()
    if square:
        gold = gold[gold.index(b"\n") + 1:]  # the header line is the caller's
    got = engine.dist_text([i[1:] for i in ids], splits(len(enc), split), square=square, pid=pid, n_slots=3)
    assert got == gold


def test_adversarial_zero_lcs_and_quirks(engine):
    """lcs == 0 -> the saturated 38-character value; orientation-dependent LCS values; single rows as blocks."""
    ids, enc = load_set(os.path.join(G, "adversarial.fasta"))
    engine.upload_seqs(enc)
    names = [i[1:] for i in ids]
    gold = open(os.path.join(G, "adversarial_dist_sq.csv"), "rb").read()
    gold = gold[gold.index(b"\n") + 1:]
    assert b"9223372036854775808.223372036854775808" in gold
    assert engine.dist_text(names, splits(len(enc), "rows"), square=True, n_slots=2) == gold
    assert engine.dist_text(names, splits(len(enc), "ragged"), pid=True, n_slots=1) == open(os.path.join(G, "adversarial_pid.csv"), "rb").read()


def edge_set():
    """Pairs (lcs l, indel) whose value sits on an edge of Double2PChar: the float just below 1 that prints as 0.000000
    (b = 2000000 loses its leading digit), integer parts of 2 and 3 digits, lcs == 0, identical sequences (0.000000)."""
    A, C_, D, X = 0, 4, 3, 22
    seqs, names = [], []

    def pair(l, indel, tag):
        x = indel // 2
        seqs.append(np.array([A] * l + [C_] * (indel - x), np.uint8))
        seqs.append(np.array([A] * l + [D] * x, np.uint8))
        names.extend([f"{tag}_ref l={l}", f"{tag}_partner|indel={indel}"])

    pair(728, 6549, "below_one")
    pair(730, 6573, "below_one_b")
    pair(1, 798, "three_digits")
    pair(2, 200, "two_digits")
    pair(300, 0, "identical")
    seqs.append(np.array([X] * 50, np.uint8))
    names.append("all_X " + "long name " * 40)  # an id longer than a workgroup
    seqs.append(np.array([A], np.uint8))
    names.append("")  # an empty id
    rng = np.random.Generator(np.random.PCG64(11))
    for k in range(40):
        seqs.append(rng.integers(0, 24, size=int(rng.integers(1, 700))).astype(np.uint8))
        names.append(f"r{k}")
    return names, seqs


@pytest.mark.parametrize("kind", [1, 0])
@pytest.mark.parametrize("square", [False, True])
def test_edge_values_against_the_oracle_format(engine, oracle, kind, square):
    names, seqs = edge_set()
    with np.errstate(over="ignore"):
        v = float(np.float32(oracle.lib.oracle_dist_indel075_f64(728, 728 + 3275, 728 + 3274)))
    assert v < 1.0 and oracle.format_dist(v) == "0.000000"  # the edge is really there
    engine.upload_seqs(seqs)
    rows = expected_rows(oracle, names, seqs, square, False, kind)
    want = b"".join(rows)
    if kind == 1:
        assert b"0.000000" in rows[1] and b"150." in rows[5] and b"9223372036854775808.223372036854775808" in want
    for split in ("one", "ragged", "rows"):
        assert engine.dist_text(names, splits(len(seqs), split), kind=kind, square=square, n_slots=3) == want, split


def test_pid_edge_values(engine, oracle):
    names, seqs = edge_set()
    engine.upload_seqs(seqs)
    want = b"".join(expected_rows(oracle, names, seqs, True, True))
    assert b"1.000000" in want and b"0.000000" in want
    assert engine.dist_text(names, splits(len(seqs), "ragged"), square=True, pid=True) == want


def test_rows_of_several_segments_and_a_partial_range(engine, oracle, tmp_path):
    """n > 1024: a row is several workgroups' work; expectation = the host writer over the oracle's matrix (the host
    formatter is pinned to oracle_format_dist and the goldens by the CPU suite)."""
    from famsa_amd import hostlib
    rng = np.random.Generator(np.random.PCG64(5))
    n = 2300
    seqs = [rng.integers(0, 20, size=int(l)).astype(np.uint8) for l in rng.integers(20, 90, size=n)]
    alphabet = "ARNDCQEGHILKMFPSTWYV"
    fasta = tmp_path / "set.fasta"
    names = [f"s{i}|{'x' * (i % 7)}" for i in range(n)]
    fasta.write_text("".join(f">{names[i]}\n{''.join(alphabet[c] for c in seqs[i])}\n" for i in range(n)))
    codes, offsets = seqio.pack(seqs)
    sq = oracle.rect(codes, offsets, np.arange(n), np.arange(n)).astype(np.uint32)
    host = hostlib.Host()
    engine.upload_seqs(seqs)
    for square in (False, True):
        path = tmp_path / f"want_{square}.csv"
        host.dist_export_from_matrix(str(fasta), sq, str(path), square_matrix=square)
        want = path.read_bytes()
        if square:
            want = want[want.index(b"\n") + 1:]
        got = engine.dist_text(names, splits(n, "ragged"), square=square, n_slots=2)
        assert got == want
    # a range in the middle only
    lines = want.split(b"\n")
    part = engine.dist_text(names, [(1000, 1100), (1100, 1101)], square=True)
    assert part == b"\n".join(lines[1000:1101]) + b"\n"


def test_wide_values(engine, oracle):
    """A sequence longer than 65535 residues: 32-bit LCS values on the way to the text."""
    rng = np.random.Generator(np.random.PCG64(8))
    seqs = [rng.integers(0, 20, size=l).astype(np.uint8) for l in (70000, 66000, 300, 1)]
    names = ["a", "b", "c", "d"]
    engine.upload_seqs(seqs)
    want = b"".join(expected_rows(oracle, names, seqs, True, False))
    assert engine.dist_text(names, [(0, 2), (2, 4)], square=True) == want


def test_misuse_is_reported(engine):
    lib, ctx = engine._lib, engine._ctx
    engine.upload_seqs([np.zeros(5, np.uint8)] * 3)
    assert lib.lcsgpu_dist_text_submit(ctx, 0, 0, 1) == -5  # before begin: LCSGPU_E_STATE
    off = np.zeros(4, np.uint64)
    assert lib.lcsgpu_dist_text_begin(ctx, b"", off.ctypes.data, 1, 0, 2) == 0
    assert lib.lcsgpu_dist_text_submit(ctx, 2, 0, 1) == -1 and lib.lcsgpu_dist_text_submit(ctx, 0, 2, 5) == -1
    assert lib.lcsgpu_dist_text_submit(ctx, 0, 0, 3) == 0
    assert lib.lcsgpu_dist_text_submit(ctx, 0, 0, 3) == -5  # the slot holds a block
    assert lib.lcsgpu_dist_text_end(ctx) == 0
