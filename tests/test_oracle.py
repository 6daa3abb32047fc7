"""CPU tests: pin the oracle (oracle/lcs_oracle.c) against the reference's own golden files and
against outputs of the reference itself (fixtures made by oracle/make_golden.py)."""
import json
import hashlib
import os

import numpy as np
import pytest

import oracle_bind
from famsa_amd import seqio

G = oracle_bind.GOLDEN


def load_set(oracle, path):
    ids, seqs = seqio.read_fasta(path)
    enc = [oracle.encode(s) for s in seqs]
    return ids, enc


def test_encode_alphabet(oracle):
    assert list(oracle.encode(seqio.ALPHABET)) == list(range(24))
    assert list(oracle.encode("arnd-c")) == [0, 1, 2, 3, 4]
    assert list(oracle.encode("JOU?1~[")) == [22] * 7
    # '{' (0x7b) - 32 = '[' -> unknown; 'x' -> 'X' = 22; 'b' -> 20
    assert list(oracle.encode("xb{")) == [22, 20, 22]


def test_quirk_values(oracle):
    """SURVEY note Q: the carry rule makes the score orientation dependent."""
    A = lambda n: np.zeros(n, np.uint8)  # noqa: E731
    assert oracle.lcs(A(192), A(1)) == 2
    assert oracle.lcs(A(192), A(2)) == 4
    assert oracle.lcs(A(192), A(9)) == 18
    assert oracle.lcs(A(1), A(192)) == 1
    assert oracle.lcs_dp(A(192), A(9)) == 9


def test_matches_plain_dp_on_random(oracle):
    rng = np.random.Generator(np.random.PCG64(7))
    for _ in range(200):
        la, lb = rng.integers(1, 300, size=2)
        k = int(rng.integers(2, 25))
        a = rng.integers(0, k, size=la).astype(np.uint8)
        b = rng.integers(0, k, size=lb).astype(np.uint8)
        assert oracle.lcs(a, b) == oracle.lcs_dp(a, b)
        assert oracle.lcs(b, a) == oracle.lcs_dp(a, b)


def test_empty_and_tiny(oracle):
    e = np.zeros(0, np.uint8)
    a = np.array([3], np.uint8)
    assert oracle.lcs(e, a) == 0
    assert oracle.lcs(a, e) == 0
    assert oracle.lcs(a, a) == 1
    assert oracle.lcs(np.array([22], np.uint8), np.array([22], np.uint8)) == 0  # X never matches
    assert oracle.lcs(np.array([21], np.uint8), np.array([21], np.uint8)) == 0  # nor Z


def test_adeno_square_vs_reference(oracle):
    ids, enc = load_set(oracle, os.path.join(G, "adeno_fiber", "adeno_fiber"))
    gold = np.load(os.path.join(G, "adeno_fiber", "lcs_square.npz"))["lcs"]
    codes, offsets = seqio.pack(enc)
    n = len(enc)
    got = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    assert got.shape == gold.shape
    assert (got == gold).all()


def test_adversarial_vs_reference(oracle):
    ids, enc = load_set(oracle, os.path.join(G, "adversarial.fasta"))
    z = np.load(os.path.join(G, "adversarial_lcs.npz"))
    codes, offsets = seqio.pack(enc)
    n = len(enc)
    got = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    assert (got == z["classic"]).all()
    assert (got == z["avx2"]).all()
    # the set does exercise the orientation dependence
    assert (got != got.T).any()


def test_long_sequences_vs_reference(oracle):
    """> 2048 residues: the reference's LoopCalculate path (lcs/lcsbp_classic.cpp:83-84)."""
    ids, enc = load_set(oracle, os.path.join(G, "adversarial_long.fasta"))
    z = np.load(os.path.join(G, "adversarial_long_lcs.npz"))
    codes, offsets = seqio.pack(enc)
    n = len(enc)
    got = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    assert (got == z["classic"]).all() and (got == z["avx2"]).all()
    assert (got != got.T).any()


def test_hemopexin_rows_and_triangle_checksum(oracle):
    ids, enc = load_set(oracle, os.path.join(G, "hemopexin", "hemopexin"))
    z = np.load(os.path.join(G, "hemopexin", "lcs_rows.npz"))
    codes, offsets = seqio.pack(enc)
    n = len(enc)
    got = oracle.rect(codes, offsets, z["rows"], np.arange(n))
    assert (got == z["lcs"]).all()
    meta = json.load(open(os.path.join(G, "meta.json")))
    tri = oracle.triangle(codes, offsets).astype(np.uint16)
    assert hashlib.sha256(tri.tobytes()).hexdigest() == meta["hemopexin"]["triangle_u16_sha256"]


def csv_from_oracle(oracle, ids, enc, square, pid):
    """The -dist_export writer (reference tree/DistanceCalculator.cpp:11-122) over oracle values."""
    codes, offsets = seqio.pack(enc)
    n = len(enc)
    lens = [len(e) for e in enc]
    lcs = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    out = []
    if square:
        out.append("".join("," + i[1:] for i in ids) + "\n")
    for i in range(n):
        row = [ids[i][1:]]
        for j in range(n if square else i):
            if pid:
                v = float(oracle.lib.oracle_pid_f32(int(lcs[i, j]), lens[i], lens[j]))
            else:
                v = float(np.float32(oracle.lib.oracle_dist_indel075_f64(int(lcs[i, j]), lens[i], lens[j])))
            row.append(oracle.format_dist(v))
        out.append(",".join(row) + "\n")
    return "".join(out).encode("latin-1")


@pytest.mark.parametrize("name,square,pid", [("dist", False, False), ("pid", False, True),
                                             ("dist_sq", True, False), ("pid_sq", True, True)])
def test_adeno_reference_csv_goldens(oracle, name, square, pid):
    """The reference's own golden files (test/adeno_fiber/*.csv) pin LCS + Transform + number format."""
    ids, enc = load_set(oracle, os.path.join(G, "adeno_fiber", "adeno_fiber"))
    got = csv_from_oracle(oracle, ids, enc, square, pid)
    gold = open(os.path.join(G, "adeno_fiber", name + ".csv"), "rb").read()
    assert got == gold


def test_adversarial_csv_with_zero_lcs(oracle):
    """lcs == 0 -> nextafter(max, 0) and the reference's integer formatting of it."""
    ids, enc = load_set(oracle, os.path.join(G, "adversarial.fasta"))
    assert csv_from_oracle(oracle, ids, enc, True, False) == open(os.path.join(G, "adversarial_dist_sq.csv"), "rb").read()
    assert csv_from_oracle(oracle, ids, enc, False, True) == open(os.path.join(G, "adversarial_pid.csv"), "rb").read()


def test_synth_generator_is_pinned(oracle):
    meta = json.load(open(os.path.join(G, "meta.json")))
    codes, offsets = seqio.synth_uniform(2000, 400)
    assert hashlib.sha256(codes.tobytes()).hexdigest() == meta["synth2k"]["codes_sha256"]
    tri = oracle.triangle(codes, offsets).astype(np.uint16)
    assert hashlib.sha256(tri.tobytes()).hexdigest() == meta["synth2k"]["triangle_u16_sha256"]


@pytest.mark.skipif(not oracle_bind.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_reference_library_still_reproduces_upstream_goldens():
    ref = oracle_bind.Ref()
    h = ref.open_fasta(os.path.join(G, "adeno_fiber", "adeno_fiber"))
    for gt in ("sl", "slink", "upgma"):
        assert ref.tree(h, gt) == open(os.path.join(G, "adeno_fiber", gt + ".dnd"), "rb").read()
    ref.close(h)
