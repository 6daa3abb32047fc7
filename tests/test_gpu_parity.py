"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C-ABI against the
committed golden fixtures (made from the reference itself) and against the oracle on the same
seeded inputs.  Integer results: the bar is bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_bind
from famsa_amd import lcsgpu, seqio

pytestmark = pytest.mark.gpu
G = oracle_bind.GOLDEN
ROOT = oracle_bind.ROOT


def load_set(path):
    ids, seqs = seqio.read_fasta(path)
    return ids, [lcsgpu.encode(s) for s in seqs]


def test_native_library_is_loaded(engine):
    import famsa_amd
    assert os.path.exists(famsa_amd.library_path())
    assert b"gfx950" in famsa_amd.load_library().lcsgpu_version()


def test_encode_matches_oracle(oracle):
    s = "ARNDCQEGHILKMFPSTWYVBZX*arndbzx-?JOU{}~ \t1"
    assert list(lcsgpu.encode(s)) == list(oracle.encode(s))


def test_adeno_square_vs_reference_golden(engine):
    ids, enc = load_set(os.path.join(G, "adeno_fiber", "adeno_fiber"))
    engine.upload_seqs(enc)
    n = len(enc)
    gold = np.load(os.path.join(G, "adeno_fiber", "lcs_square.npz"))["lcs"]
    got = engine.lcs_rect((0, n), (0, n))
    assert got.dtype == np.uint16 and got.shape == (n, n)
    assert (got == gold).all()
    got32 = engine.lcs_rect(np.arange(n), np.arange(n), dtype=np.uint32)
    assert (got32 == gold).all()


def test_adversarial_quirk_and_edges_vs_reference_golden(engine):
    """Homopolymer carry quirk (orientation dependent), lengths 1..2048, B/Z/X/*, duplicates."""
    ids, enc = load_set(os.path.join(G, "adversarial.fasta"))
    engine.upload_seqs(enc)
    n = len(enc)
    gold = np.load(os.path.join(G, "adversarial_lcs.npz"))["classic"]
    got = engine.lcs_rect((0, n), (0, n))
    bad = np.argwhere(got != gold)
    assert len(bad) == 0, f"first mismatches (ref,partner): {bad[:10].tolist()}"
    assert (got != got.T).any()  # the orientation dependence is really exercised
    tri = engine.lcs_triangle()
    il = np.tril_indices(n, -1)
    assert (tri == gold[il]).all()


def test_hemopexin_triangle_checksum_and_rows(engine, oracle):
    ids, enc = load_set(os.path.join(G, "hemopexin", "hemopexin"))
    engine.upload_seqs(enc)
    n = len(enc)
    meta = json.load(open(os.path.join(G, "meta.json")))
    tri = engine.lcs_triangle()
    assert hashlib.sha256(tri.tobytes()).hexdigest() == meta["hemopexin"]["triangle_u16_sha256"]
    z = np.load(os.path.join(G, "hemopexin", "lcs_rows.npz"))
    got = engine.lcs_rect(z["rows"], (0, n))
    assert (got == z["lcs"]).all()
    # row-block split (what each rank of a multi-GPU job computes) is the same triangle
    cuts = [0, 1, 700, 701, 2048, n]
    parts = [engine.lcs_triangle(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    assert (np.concatenate(parts) == tri).all()


def test_sorted_order_triangle_vs_oracle(engine, oracle):
    """The tree builders run on the length-sorted set (reference msa.cpp:245-279)."""
    ids, enc = load_set(os.path.join(G, "hemopexin", "hemopexin"))
    order = seqio.sort_order(enc)[:1500]
    enc = [enc[i] for i in order]
    engine.upload_seqs(enc)
    codes, offsets = seqio.pack(enc)
    want = oracle.triangle(codes, offsets)
    got = engine.lcs_triangle(dtype=np.uint32)
    assert (got == want).all()


def test_synthetic_2k_checksum(engine):
    meta = json.load(open(os.path.join(G, "meta.json")))
    codes, offsets = seqio.synth_uniform(2000, 400)
    engine.upload(codes, offsets)
    tri = engine.lcs_triangle()
    assert hashlib.sha256(tri.tobytes()).hexdigest() == meta["synth2k"]["triangle_u16_sha256"]


def test_gather_lists_ragged_shapes(engine, oracle):
    """calculateDistanceRange-style calls: one or a few refs against an arbitrary id list."""
    rng = np.random.Generator(np.random.PCG64(11))
    seqs = [rng.integers(0, 24, size=int(l)).astype(np.uint8) for l in rng.integers(0, 700, size=333)]
    seqs[5] = np.zeros(0, np.uint8)  # empty sequence
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    for n_refs, n_cols in [(1, 1), (1, 257), (3, 64), (17, 300), (40, 333), (5, 0), (0, 5)]:
        refs = rng.integers(0, len(seqs), size=n_refs)
        cols = rng.integers(0, len(seqs), size=n_cols)
        got = engine.lcs_rect(refs, cols, dtype=np.uint32)
        want = oracle.rect(codes, offsets, refs, cols) if n_refs and n_cols else np.zeros((n_refs, n_cols), np.uint32)
        assert got.shape == want.shape
        assert (got == want).all()


def test_all_word_counts(engine, oracle):
    """Every instantiated word count 1..32 (reference unrolls 1..32, lcs/lcsbp_classic.cpp:47-85)."""
    rng = np.random.Generator(np.random.PCG64(5))
    lens = [1 + 64 * k + int(rng.integers(0, 64)) for k in range(32)] + [64 * k for k in range(1, 33)]
    seqs = [rng.integers(0, 20, size=l).astype(np.uint8) for l in lens]
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    want = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    got = engine.lcs_rect((0, n), (0, n))
    bad = np.argwhere(got != want)
    assert len(bad) == 0, bad[:10].tolist()


def test_long_sequences_vs_reference_golden(engine):
    """Refs > 2048 residues go through the segmented kernel with carry streams."""
    ids, enc = load_set(os.path.join(G, "adversarial_long.fasta"))
    engine.upload_seqs(enc)
    n = len(enc)
    gold = np.load(os.path.join(G, "adversarial_long_lcs.npz"))["classic"]
    got = engine.lcs_rect((0, n), (0, n))
    bad = np.argwhere(got != gold)
    assert len(bad) == 0, f"(ref,partner) mismatches: {bad[:10].tolist()}"
    tri = engine.lcs_triangle()
    assert (tri == gold[np.tril_indices(n, -1)]).all()


def test_long_random_vs_oracle(engine, oracle):
    rng = np.random.Generator(np.random.PCG64(21))
    lens = [2049, 2500, 4096, 7000, 9000, 300, 64, 2048] + [int(x) for x in rng.integers(1, 6000, size=300)]
    seqs = [rng.integers(0, 22, size=l).astype(np.uint8) for l in lens]
    seqs[4][2048 + 64 * 3: 2048 + 64 * 5] = 7  # quirk words in the second segment of a long ref
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    refs = np.arange(0, n, 3)
    got = engine.lcs_rect(refs, (0, n), dtype=np.uint32)
    want = oracle.rect(codes, offsets, refs, np.arange(n))
    bad = np.argwhere(got != want)
    assert len(bad) == 0, bad[:10].tolist()


def test_every_halfword_count(engine, oracle):
    """Every instantiated half-word count (lengths 1..2048 in steps of 32, both edges)."""
    rng = np.random.Generator(np.random.PCG64(9))
    lens = sorted(set([32 * k for k in range(1, 65)] + [32 * k + 1 for k in range(0, 64)] + [31, 33, 2047]))
    seqs = [rng.integers(0, 20, size=l).astype(np.uint8) for l in lens]
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    cols = np.arange(0, n, 5)
    want = oracle.rect(codes, offsets, np.arange(n), cols)
    got = engine.lcs_rect((0, n), cols)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, bad[:10].tolist()


def _prim_reference(oracle, codes, offsets, kind_fn):
    """MSTPrim::run_view's recurrence (reference tree/MSTPrim.cpp:356-533) in plain Python over oracle LCS."""
    n = len(offsets) - 1
    lens = np.diff(offsets.astype(np.int64))
    sq = oracle.rect(codes, offsets, np.arange(n), np.arange(n))  # sq[ref, partner]
    key = [(np.finfo(np.float64).max, 0)] * n
    alive = set(range(1, n))
    cur, edges = 0, []
    M = (1 << 64) - 1
    while alive:
        for v in alive:
            d = kind_fn(int(sq[cur, v]), int(lens[cur]), int(lens[v]))
            if d <= key[v][0]:
                a, b = min(cur, v), max(cur, v)
                s = (d, M ^ ((a << 32) + b))
                if s < key[v]:
                    key[v] = s
        best = min(alive, key=lambda v: key[v])
        packed = M ^ key[best][1]
        edges.append((packed >> 32, packed & 0xffffffff, key[best][0]))
        alive.remove(best)
        cur = best
    return edges


def test_device_prim_edges_vs_reference_recurrence(engine, oracle):
    ids, enc = load_set(os.path.join(G, "adversarial_tree.fasta"))  # includes orientation-sensitive refs
    enc = [enc[i] for i in seqio.sort_order(enc)]
    rng = np.random.Generator(np.random.PCG64(3))
    enc += [rng.integers(0, 20, size=int(l)).astype(np.uint8) for l in rng.integers(40, 300, size=500)]
    engine.upload_seqs(enc)
    assert engine.orientation_flags().sum() >= 2
    codes, offsets = seqio.pack(enc)
    for kind, fn in [(1, oracle.lib.oracle_dist_indel075_f64), (0, oracle.lib.oracle_dist_indel_f64)]:
        got = engine.mst_prim(kind)
        want = _prim_reference(oracle, codes, offsets, fn)
        assert [(int(e["from"]), int(e["to"])) for e in got] == [(a, b) for a, b, _ in want]
        assert [float(e["dist"]) for e in got] == [d for _, _, d in want]


def test_errors_are_reported(engine):
    import famsa_amd
    engine.upload_seqs([np.zeros(10, np.uint8)] * 3)
    with pytest.raises(famsa_amd.LcsGpuError):
        engine.lcs_rect(np.array([7]), (0, 3))
    with pytest.raises(famsa_amd.LcsGpuError):
        engine.lcs_rect((0, 3), (1, 3))
    with pytest.raises(famsa_amd.LcsGpuError):
        engine.upload_seqs([np.full(4, 40, np.uint8)])


def test_row_minima_vs_oracle(engine, oracle):
    """Per-row nearest neighbour with MSTPrim's tie rule, distances as Transform<double,...>."""
    import torch
    ids, enc = load_set(os.path.join(G, "hemopexin", "hemopexin"))
    order = seqio.sort_order(enc)[:1200]
    enc = [enc[i] for i in order] + [np.full(40, 22, np.uint8)]  # last row: lcs 0 against everything
    engine.upload_seqs(enc)
    n = len(enc)
    lens = np.array([len(e) for e in enc])
    codes, offsets = seqio.pack(enc)
    lcs = oracle.triangle(codes, offsets)
    for kind, fn in [(1, oracle.lib.oracle_dist_indel075_f64), (0, oracle.lib.oracle_dist_indel_f64)]:
        r0, r1 = 0, n
        pairs = n * (n - 1) // 2
        tri = torch.empty(pairs, dtype=torch.int16, device="cuda:0")
        engine.lcs_triangle_dev(r0, r1, tri.data_ptr(), 2)
        out = torch.zeros((n, 2), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()  # torch fills on ITS stream; the engine writes `out` on its own
        engine.row_minima_dev(tri.data_ptr(), 2, r0, r1, kind, out.data_ptr(), sync=True)
        d = out[:, 0].cpu().numpy()
        j = out[:, 1].cpu().numpy().view(np.int64)
        assert j[0] == -1 and d[0] == np.finfo(np.float64).max
        for i in list(range(1, 40)) + list(range(40, n, 37)) + [n - 1]:
            row = lcs[i * (i - 1) // 2: i * (i - 1) // 2 + i]
            dd = np.array([fn(int(l), int(lens[i]), int(lens[k])) for k, l in enumerate(row)])
            m = dd.min()
            want_j = int(np.max(np.nonzero(dd == m)[0]))
            assert d[i] == m and j[i] == want_j, (i, d[i], m, j[i], want_j)


def test_triangle_over_id_lists(engine, oracle):
    """calculateDistanceMatrix on a subset (FastTree sample matrix / per-cluster matrices)."""
    rng = np.random.Generator(np.random.PCG64(17))
    lens = [int(x) for x in rng.integers(1, 900, size=700)] + [2300, 2600, 64 * 3, 64 * 5]
    seqs = [rng.integers(0, 21, size=l).astype(np.uint8) for l in lens]
    seqs[-2][64:128] = 3   # orientation-sensitive members
    seqs[-1][128:256] = 9
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    for m in (1, 2, 5, 300, n):
        ids = rng.permutation(n)[:m]
        if m == n:
            ids = np.array(seqio.sort_order(seqs))  # length-sorted, the usual case
        got = engine.lcs_triangle_ids(ids, dtype=np.uint32)
        sq = oracle.rect(codes, offsets, ids, ids)
        want = sq[np.tril_indices(m, -1)]
        assert (got == want).all(), m


def _random_groups(rng, n, sizes):
    return [rng.permutation(n)[:m].astype(np.int32) for m in sizes]


def test_triangles_batch_vs_oracle(engine, oracle):
    """The leaf matrices of one FastTree split in one call: every list's triangle, lists of 0 / 1 / 2
    members, word-count classes mixed inside a list, orientation-sensitive members."""
    rng = np.random.Generator(np.random.PCG64(23))
    lens = [int(x) for x in rng.integers(1, 700, size=900)] + [64 * 3, 64 * 5, 64 * 2, 2048]
    seqs = [rng.integers(0, 21, size=l).astype(np.uint8) for l in lens]
    seqs[-4][64:128] = 3   # orientation-sensitive members
    seqs[-3][128:256] = 9
    seqs[-2][64:128] = 0
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    groups = _random_groups(rng, n, [0, 1, 2, 3, 40, 257, 1, 300, 0, 77, 513, 5])
    groups.append(np.array([n - 4, 5, n - 3, n - 2, 9, n - 1, n - 4], np.int32))  # quirk refs and a repeated id
    groups.append(np.array(seqio.sort_order(seqs)[:350], np.int32))
    for dtype in (np.uint16, np.uint32):
        got = engine.lcs_triangles_batch(groups, dtype=dtype)
        assert len(got) == len(groups)
        for g, ids in enumerate(groups):
            m = len(ids)
            want = oracle.rect(codes, offsets, ids, ids)[np.tril_indices(m, -1)] if m > 1 else np.zeros(0)
            assert got[g].shape == (m * (m - 1) // 2,)
            assert (got[g] == want).all(), (g, m)
        one_by_one = [engine.lcs_triangle_ids(ids, dtype=dtype) for ids in groups if len(ids) > 1]
        assert all((a == b).all() for a, b in zip([x for x in got if len(x)], [x for x in one_by_one if len(x)]))


def test_triangles_batch_with_long_members(engine, oracle):
    """Lists containing members beyond 2048 residues take the list-by-list path; same values."""
    rng = np.random.Generator(np.random.PCG64(29))
    lens = [int(x) for x in rng.integers(20, 400, size=60)] + [2100, 2500, 3000]
    seqs = [rng.integers(0, 20, size=l).astype(np.uint8) for l in lens]
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    groups = [np.arange(n, dtype=np.int32)[::-1].copy(), np.array([n - 1, 3, n - 2], np.int32), np.array([4], np.int32)]
    got = engine.lcs_triangles_batch(groups, dtype=np.uint32)
    for g, ids in enumerate(groups):
        m = len(ids)
        want = oracle.rect(codes, offsets, ids, ids)[np.tril_indices(m, -1)] if m > 1 else np.zeros(0)
        assert (got[g] == want).all(), g


def test_concurrent_host_threads_share_one_context(engine, oracle):
    """The reference runs one CLCSBP per worker thread; here many threads share the engine and their
    host-memory calls run on separate lanes (streams).  Every thread must get its own results."""
    import threading
    rng = np.random.Generator(np.random.PCG64(101))
    seqs = [rng.integers(0, 20, size=int(l)).astype(np.uint8) for l in rng.integers(20, 700, size=900)]
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    jobs = []
    for t in range(24):
        r = np.random.Generator(np.random.PCG64(t))
        jobs.append((r.integers(0, n, size=int(r.integers(1, 40))), r.integers(0, n, size=int(r.integers(1, 600))),
                     r.permutation(n)[: int(r.integers(2, 200))]))
    results = [None] * len(jobs)
    errors = []

    def work(k):
        try:
            out = []
            for rep in range(5):
                refs, cols, ids = jobs[k]
                out.append((engine.lcs_rect(refs, cols, dtype=np.uint32), engine.lcs_triangle_ids(ids, dtype=np.uint32)))
            results[k] = out
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k, (refs, cols, ids) in enumerate(jobs):
        want_r = oracle.rect(codes, offsets, refs, cols)
        sq = oracle.rect(codes, offsets, ids, ids)
        want_t = sq[np.tril_indices(len(ids), -1)]
        for got_r, got_t in results[k]:
            assert (got_r == want_r).all() and (got_t == want_t).all(), k


def test_degenerate_set_sizes(engine, oracle):
    """0, 1 and 2 sequences; empty members; a call before any work exists."""
    engine.upload_seqs([])
    assert engine.lcs_triangle().size == 0
    engine.upload_seqs([np.array([1, 2, 3], np.uint8)])
    assert engine.lcs_triangle().size == 0
    assert engine.lcs_rect((0, 1), (0, 1))[0, 0] == 3
    assert len(engine.mst_prim(1)) == 0
    a, b = np.array([0, 1, 2, 3, 4], np.uint8), np.array([9, 1, 2, 9], np.uint8)
    engine.upload_seqs([a, b, np.zeros(0, np.uint8)])
    tri = engine.lcs_triangle(dtype=np.uint32)
    assert list(tri) == [oracle.lcs(b, a), 0, 0]
    e = engine.mst_prim(1)
    assert len(e) == 2 and {(int(x["from"]), int(x["to"])) for x in e} <= {(0, 1), (0, 2), (1, 2)}


def test_sets_with_a_sequence_beyond_65535_residues(engine, oracle):
    """One member longer than 65535 residues switches every result to 32-bit elements: the same calls,
    the uint32 instantiations of the consumers (batched triangles, seed assignment, CLARANS distances)."""
    import famsa_amd
    rng = np.random.Generator(np.random.PCG64(31))
    lens = [int(x) for x in rng.integers(30, 260, size=40)] + [66000]
    seqs = [rng.integers(0, 20, size=l).astype(np.uint8) for l in lens]
    seqs[-1][: lens[3]] = seqs[3]  # give the giant something in common with a short one
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    ids = np.arange(n, dtype=np.int32)
    want = oracle.rect(codes, offsets, ids, ids)
    got = engine.lcs_rect(ids, ids, dtype=np.uint32)
    assert (got == want).all()
    assert want[n - 1, n - 1] == 66000
    with pytest.raises(famsa_amd.LcsGpuError):
        engine.lcs_rect(ids, ids, dtype=np.uint16)  # 16-bit results cannot hold this set's values
    groups = [np.array([n - 1, 3, 7, 1], np.int32), np.arange(10, dtype=np.int32), np.array([5, n - 1], np.int32)]
    tri = engine.lcs_triangles_batch(groups, dtype=np.uint32)
    for g, gi in zip(tri, groups):
        assert (g == want[np.ix_(gi, gi)][np.tril_indices(len(gi), -1)]).all()
    # seed assignment and CLARANS read the same 32-bit rectangle / triangle on the device
    seeds = np.array([n - 1, 4, 9], np.int32)
    fn = oracle.lib.oracle_dist_indel075_f32
    d0 = np.array([fn(int(want[0, c]), lens[0], lens[c]) for c in ids], np.float32)
    wd, wa = d0.copy(), np.zeros(n, np.int32)
    for r, s in enumerate(seeds):
        for c in ids:
            d = np.float32(fn(int(want[s, c]), lens[s], lens[c]))
            if d < wd[c]:
                wd[c], wa[c] = d, 1 + r
    gd, ga = d0.copy(), np.zeros(n, np.int32)
    engine.assign_seeds(seeds, ids, gd, ga)
    assert (ga == wa).all() and (gd.view(np.uint32) == wd.view(np.uint32)).all()
    from famsa_amd import hostlib as host_bind
    sub = np.array([n - 1] + list(range(30)), np.int32)
    lcs_tri = want[np.ix_(sub, sub)][np.tril_indices(len(sub), -1)]
    dist = oracle.dist_triangle_f32(lcs_tri, np.array([lens[i] for i in sub], np.uint32), 1)
    assert engine.clarans(sub, 4).tolist() == host_bind.Host().clarans(dist, len(sub), 4).tolist()


@pytest.mark.parametrize("shape", ["ties", "family", "tiny", "prefixes", "prefixes-sorted", "family-sorted", "two-lengths"])
def test_boruvka_mst_equals_the_prim_kernel(engine, monkeypatch, shape):
    """lcsgpu_mst_prim builds the tree by Boruvka rounds when distances are orientation free and orders the
    edges by a Prim walk over the tree: same edges, same order, same distances as the step-by-step Prim
    kernel (which the oracle's recurrence pins, test_device_prim_edges_vs_reference_recurrence)."""
    rng = np.random.Generator(np.random.PCG64(53))
    if shape == "ties":
        seqs = [rng.integers(0, 3, size=int(rng.integers(3, 12))).astype(np.uint8) for _ in range(1500)]
    elif shape == "family":
        anc = rng.integers(0, 20, size=200, dtype=np.uint8)
        seqs = []
        for _ in range(3000):
            s = anc.copy()
            m = rng.random(200) < 0.2
            s[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
            seqs.append(s[: int(rng.integers(120, 201))].copy())
    elif shape.startswith("prefixes"):
        # prefixes of one ancestor: LCS = the shorter length, so equal distances come from many different
        # (LCS, lengths) combinations and near-equal ones from neighbouring ones -- what the passes' integer
        # pre-filter (mst_kernels.hip, l_threshold) must not cut; unsorted = wide length spread inside a batch
        anc = rng.integers(0, 20, size=420, dtype=np.uint8)
        seqs = [anc[: int(l)].copy() for l in rng.integers(40, 421, size=2600)]
        if shape.endswith("sorted"):
            seqs.sort(key=lambda q: -len(q))
    elif shape == "family-sorted":
        anc = rng.integers(0, 20, size=300, dtype=np.uint8)
        seqs = []
        for _ in range(2600):
            q = anc.copy()
            m = rng.random(300) < 0.05
            q[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
            seqs.append(q[: int(rng.integers(200, 301))].copy())
        seqs.sort(key=lambda q: -len(q))
    elif shape == "two-lengths":
        # blocks of long sequences interleaved with short ones: the block minima of the filter are far below
        # the best edge's length in every batch
        seqs = [rng.integers(0, 4, size=(300 if (i // 7) % 2 else 30)).astype(np.uint8) for i in range(2100)]
    else:
        seqs = [rng.integers(0, 20, size=int(rng.integers(5, 40))).astype(np.uint8) for _ in range(3)]
    engine.upload_seqs(seqs)
    assert engine.orientation_flags().sum() == 0
    for kind in (0, 1, 1 | 0x100):
        monkeypatch.delenv("LCSGPU_MST_MODE", raising=False)
        fast = engine.mst_prim(kind)
        monkeypatch.setenv("LCSGPU_MST_MODE", "prim")
        slow = engine.mst_prim(kind)
        monkeypatch.delenv("LCSGPU_MST_MODE", raising=False)
        assert (fast["from"] == slow["from"]).all() and (fast["to"] == slow["to"]).all()
        assert (fast["dist"].view(np.uint64) == slow["dist"].view(np.uint64)).all()


@pytest.mark.parametrize("batch", ["32", "0"])
def test_upgma_without_a_finite_neighbour_is_an_error(engine, monkeypatch, batch):
    """A sequence that shares no residue with any other (all X: LCS 0, distance FLT_MAX >= the reference's BIG_DIST) leaves
    UPGMA::computeTree without a row to pick at the last merge -- undefined behaviour in the reference
    (tree/UPGMA.cpp:198-220 reads index 0x7FFFFFFF).  The device reducers answer LCSGPU_E_INVALID, in batches of merges
    (upgma_batch_kernels.hip: the walk meets a key >= BIG_DIST) and with one launch per merge."""
    import famsa_amd
    monkeypatch.setenv("LCSGPU_UPGMA_BATCH", batch)
    rng = np.random.Generator(np.random.PCG64(5))
    seqs = [rng.integers(0, 20, size=int(l)).astype(np.uint8) for l in rng.integers(30, 90, size=300)]
    seqs.insert(137, np.full(40, 22, np.uint8))
    engine.upload_seqs(seqs)
    with pytest.raises(famsa_amd.LcsGpuError, match="no finite nearest neighbour"):
        engine.upgma(1, False)
    engine.upload_seqs(seqs[:137] + seqs[138:])  # without it: a tree
    left, right = engine.upgma(1, False)
    assert len(left) == 299 and left.max() < 2 * 300 - 2


def test_upload_ordered_is_the_upload_of_the_packed_order(engine, oracle):
    """lcsgpu_upload_ordered gathers the caller's order on the device: the set built from (records, order) -- a
    permutation that drops records, names one twice, holds empty, quirk-sensitive and > 2048-residue members, ragged
    tiles -- answers every query like the set uploaded from the packed copy of that order: lengths, orientation
    flags, the square of oriented LCS values (against the reference's golden where one exists, the oracle elsewhere)."""
    ids, enc = load_set(os.path.join(G, "adversarial.fasta"))
    rng = np.random.Generator(np.random.PCG64(31))
    extra = [rng.integers(0, 20, size=int(L), dtype=np.uint8) for L in (0, 1, 63, 64, 65, 130, 700, 2500)]
    records = list(enc) + extra
    codes, offsets = seqio.pack(records)
    order = rng.permutation(len(records))[: len(records) - 5].astype(np.int32)
    order = np.concatenate([order, order[:3]])  # three records twice
    engine.upload_ordered(codes, offsets, order)
    n = len(order)
    assert engine.n == n
    got = engine.lcs_rect((0, n), (0, n), dtype=np.uint32)
    flags_ordered = engine.orientation_flags()
    tri_ordered = engine.lcs_triangle()
    packed = [records[i] for i in order]
    engine.upload_seqs(packed)
    assert (engine.lengths == np.array([len(r) for r in packed], np.uint32)).all()
    want = engine.lcs_rect((0, n), (0, n), dtype=np.uint32)
    assert (got == want).all()
    assert (flags_ordered == engine.orientation_flags()).all() and flags_ordered.any()
    assert (tri_ordered == engine.lcs_triangle()).all()
    pc, po = seqio.pack(packed)
    sample = rng.integers(0, n, size=40)
    assert (oracle.rect(pc, po, sample, sample) == got[np.ix_(sample, sample)]).all()
    # the errors: an entry that is no record, NULL order with a different count
    bad = order.copy()
    bad[2] = len(records)
    with pytest.raises(lcsgpu.LcsGpuError):
        engine.upload_ordered(codes, offsets, bad)
    bad[2] = -1
    with pytest.raises(lcsgpu.LcsGpuError):
        engine.upload_ordered(codes, offsets, bad)
    # an empty order: an empty set
    engine.upload_ordered(codes, offsets, np.zeros(0, np.int32))
    assert engine.n == 0
