"""Device CLARANS (lcsgpu_clarans) against the reference's CLARANS::operator() -- or, where oracle/_ref
is not built, the host search that the CPU suite pins to it -- on the oracle's float distance triangle."""
import time

import numpy as np
import pytest

from famsa_amd import hostlib as host_bind
import oracle_bind
from famsa_amd import seqio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def searcher():
    if oracle_bind.have_ref():
        return oracle_bind.Ref().clarans
    return host_bind.Host().clarans


def _family(rng, n, length, mut):
    anc = rng.integers(0, 20, size=length, dtype=np.uint8)
    out = []
    for _ in range(n):
        s = anc.copy()
        m = rng.random(length) < mut
        s[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
        out.append(s[: int(rng.integers(int(length * 0.7), length + 1))].copy())
    return out


def _short(rng, n, lo, hi, alpha):
    return [rng.integers(0, alpha, size=int(rng.integers(lo, hi + 1)), dtype=np.uint8) for _ in range(n)]


def _expected_triangle(oracle, seqs, ids, kind):
    sub = [seqs[i] for i in ids]
    codes, offsets = seqio.pack(sub)
    lcs = oracle.triangle(codes, offsets)
    lens = np.array([len(s) for s in sub], np.uint32)
    return oracle.dist_triangle_f32(lcs, lens, kind)


CASES = [
    # name, generator, n_set, n_sample, k, fixed, fraction, searches, kind
    ("family-default-shape", "family", 2300, 2000, 100, 1, 0.1, 2, 1),
    ("family-small", "family", 400, 300, 20, 1, 0.1, 2, 1),
    ("family-indel-div-lcs", "family", 400, 350, 25, 1, 0.2, 3, 0),
    ("two-registers-per-lane", "family", 800, 700, 130, 1, 0.05, 2, 1),
    ("no-fixed-medoid", "family", 300, 300, 10, 0, 0.1, 2, 1),
    ("three-fixed", "family", 300, 256, 16, 3, 0.1, 2, 1),
    ("whole-neighbourhood", "family", 200, 150, 8, 1, 1.0, 1, 1),
    ("ties-everywhere", "short", 350, 300, 20, 1, 0.1, 2, 1),
    ("all-medoids", "family", 40, 24, 24, 1, 0.1, 2, 1),
    ("one-medoid", "family", 40, 30, 1, 0, 0.1, 2, 1),
    ("tiny", "family", 8, 3, 2, 1, 0.1, 2, 1),
    ("single-member", "family", 8, 1, 1, 0, 0.1, 1, 1),
    # few medoids: a candidate is closer than their medoid to a large part of the members (every slot can go negative)
    ("one-medoid-many-members", "family", 1300, 1200, 1, 0, 0.02, 2, 1),
    ("three-medoids", "family", 2100, 2040, 3, 1, 0.01, 2, 1),
    ("most-non-medoids-the-device-takes", "family", 2200, 2088, 40, 1, 0.01, 1, 1),
    # more than 512 medoids: two slots per lane in the walk (the kernel's second instantiation)
    ("more-than-512-medoids", "family", 1500, 1400, 600, 1, 0.02, 1, 1),
    # duplicate members: distances tie exactly (d_new == dn with a smaller slot, equal second-nearest medoids)
    ("duplicates-tie-exactly", "dups", 300, 280, 12, 1, 0.3, 2, 1),
]


@pytest.mark.parametrize("name,gen,n_set,n_sample,k,fixed,frac,iters,kind", CASES, ids=[c[0] for c in CASES])
def test_device_clarans_matches_reference(engine, oracle, searcher, name, gen, n_set, n_sample, k, fixed, frac, iters,
                                          kind):
    rng = np.random.default_rng(int.from_bytes(name.encode(), "little") % (1 << 31))
    seqs = _family(rng, n_set, 180, 0.25) if gen in ("family", "dups") else _short(rng, n_set, 6, 14, 3)
    if gen == "dups":  # every sequence four times: members at distance 0 of each other, medoids at equal distances
        seqs = [seqs[i // 4] for i in range(n_set)]
    engine.upload_seqs(seqs)
    ids = np.sort(rng.permutation(n_set)[:n_sample]).astype(np.int32)
    if name == "no-fixed-medoid":
        ids = rng.permutation(n_set)[:n_sample].astype(np.int32)  # any order is a valid sample order
    tri = _expected_triangle(oracle, seqs, ids, kind)
    want = searcher(tri, n_sample, k, fixed, frac, iters)
    t0 = time.time()
    got = engine.clarans(ids, k, fixed, frac, iters, kind)
    print(f"{name}: device search {1e3 * (time.time() - t0):.1f} ms")
    assert got.tolist() == want.tolist()


def test_shapes_beyond_the_device_search_are_refused_not_approximated(engine):
    """More than 2048 non-medoids (or 1024 medoids): LCSGPU_E_UNSUPPORTED -- the host layer then searches on the host
    (famsa_amd/host/lcs_source.cpp: GpuLcsSource::clarans returns false)."""
    import famsa_amd
    rng = np.random.default_rng(6)
    engine.upload_seqs(_family(rng, 2600, 60, 0.2))
    with pytest.raises(famsa_amd.LcsGpuError) as e:
        engine.clarans(np.arange(2500, dtype=np.int32), 3, 1, 0.01, 1)
    assert "at most" in str(e.value)


def test_device_clarans_rejects_bad_shapes(engine):
    import famsa_amd
    rng = np.random.default_rng(5)
    engine.upload_seqs(_family(rng, 20, 60, 0.2))
    ids = np.arange(10, dtype=np.int32)
    for args in [(11, 1), (0, 0), (4, 4), (4, -1)]:
        with pytest.raises(famsa_amd.LcsGpuError):
            engine.clarans(ids, args[0], args[1])
    with pytest.raises(famsa_amd.LcsGpuError):
        engine.clarans(np.array([0, 25], np.int32), 1, 0)


@pytest.mark.parametrize("kind", [0, 1])
def test_assign_seeds_matches_the_host_sweep(engine, oracle, kind):
    """lcsgpu_assign_seeds against FastTree::makeEvaluation's sweep restated with the oracle's LCS and float
    transform: seeds in order, strict '<', columns given as an id list, chunk borders, ties (short sequences)."""
    rng = np.random.default_rng(77 + kind)
    seqs = _family(rng, 500, 120, 0.3) + _short(rng, 300, 4, 9, 3)
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    cols = rng.permutation(n)[:700].astype(np.int32)
    seeds = rng.permutation(n)[:13].astype(np.int32)
    lcs = oracle.rect(codes, offsets, seeds, cols)
    fn = oracle.lib.oracle_dist_indel075_f32 if kind == 1 else oracle.lib.oracle_dist_indel_f32
    lens = [len(s) for s in seqs]
    first = np.array([fn(int(oracle.lcs(seqs[cols[0]], seqs[c])), lens[cols[0]], lens[c]) for c in cols], np.float32)
    want_d, want_a = first.copy(), np.zeros(len(cols), np.int32)
    for r, sid in enumerate(seeds):
        for j, c in enumerate(cols):
            d = np.float32(fn(int(lcs[r, j]), lens[sid], lens[c]))
            if d < want_d[j]:
                want_d[j] = d
                want_a[j] = 1 + r
    got_d, got_a = first.copy(), np.zeros(len(cols), np.int32)
    engine.assign_seeds(seeds, cols, got_d, got_a, first_k=1, kind=kind)
    assert (got_a == want_a).all()
    assert (got_d.view(np.uint32) == want_d.view(np.uint32)).all()


def test_many_samples_in_one_call(engine, oracle, searcher):
    """lcsgpu_clarans_batch: samples of several shapes (some shared, one with every member a medoid, a single member) in
    one call = every sample searched alone; three of them also against the reference's own search."""
    rng = np.random.default_rng(99)
    seqs = _family(rng, 2600, 150, 0.25)
    engine.upload_seqs(seqs)
    shapes = [(2000, 100)] * 3 + [(600, 30)] * 2 + [(24, 24), (2, 2), (310, 7), (2050, 2), (1500, 600)]
    samples = [np.sort(rng.permutation(len(seqs))[:m]).astype(np.int32) for m, _ in shapes]
    ks = [k for _, k in shapes]
    got = engine.clarans_batch(samples, ks, 1, 0.1, 2)
    for i in (0, 3, 7):  # (7: 310 members, 7 medoids)
        want = searcher(_expected_triangle(oracle, seqs, samples[i], 1), len(samples[i]), ks[i], 1, 0.1, 2)
        assert got[i].tolist() == want.tolist(), i
    for i, (ids, k) in enumerate(zip(samples, ks)):
        assert got[i].tolist() == engine.clarans(ids, k, 1, 0.1, 2).tolist(), i
    # a sample outside the device search's shapes: the whole call is refused, nothing is approximated
    import famsa_amd
    with pytest.raises(famsa_amd.LcsGpuError) as e:
        engine.clarans_batch(samples + [np.arange(2500, dtype=np.int32)], ks + [3], 1, 0.1, 1)
    assert "at most" in str(e.value)


@pytest.mark.parametrize("kind", [0, 1])
def test_assign_seeds_batch_matches_the_host_sweep(engine, oracle, kind):
    """lcsgpu_assign_seeds_batch: several evaluations from scratch in one call against the sweep restated with the oracle's
    LCS and float transform (first seed's row, then strict '<' in seed order); ties on short sequences; an evaluation
    without columns; seeds of many lengths (several word-count classes in one call)."""
    rng = np.random.default_rng(177 + kind)
    seqs = _family(rng, 500, 120, 0.3) + _short(rng, 300, 4, 9, 3) + _family(rng, 60, 400, 0.2)
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    lens = [len(s) for s in seqs]
    fn = oracle.lib.oracle_dist_indel075_f32 if kind == 1 else oracle.lib.oracle_dist_indel_f32
    seeds, cols = [], []
    for m, k in [(700, 13), (1, 1), (0, 4), (300, 40), (860, 5), (90, 90)]:
        cols.append(rng.permutation(n)[:m].astype(np.int32))
        seeds.append(rng.permutation(n)[:k].astype(np.int32))
    got_d, got_a = engine.assign_seeds_batch(seeds, cols, kind=kind)
    for g, (sd, cl) in enumerate(zip(seeds, cols)):
        if len(cl) == 0:
            continue
        lcs = oracle.rect(codes, offsets, sd, cl)
        want_d = np.full(len(cl), np.inf, np.float32)
        want_a = np.zeros(len(cl), np.int32)
        for r, sid in enumerate(sd):
            d = np.array([fn(int(lcs[r, j]), lens[sid], lens[c]) for j, c in enumerate(cl)], np.float32)
            better = d < want_d
            want_d[better] = d[better]
            want_a[better] = r
        assert (got_a[g] == want_a).all(), g
        assert (got_d[g].view(np.uint32) == want_d.view(np.uint32)).all(), g


def test_concurrent_searches_share_the_batch(engine):
    """Searches of several host threads at once: every thread must get exactly what it gets when it searches alone
    (different shapes; each call runs its own launches on its own lane)."""
    import threading
    rng = np.random.default_rng(2024)
    seqs = _family(rng, 2500, 160, 0.25)
    engine.upload_seqs(seqs)
    jobs = []
    for t in range(20):  # (with clarans_groups=1 below: more than the 16 searches a look holds)
        m = int(rng.integers(150, 1400))
        ids = np.sort(rng.permutation(len(seqs))[:m]).astype(np.int32)
        jobs.append((ids, int(rng.integers(2, 40)), int(rng.integers(0, 2)), float(rng.choice([0.05, 0.1, 0.3])),
                     int(rng.integers(1, 4))))
    alone = [engine.clarans(ids, k, fixed, frac, iters).tolist() for ids, k, fixed, frac, iters in jobs]
    for attempt in range(2):
        together = [None] * len(jobs)
        errors = []

        def work(i):
            try:
                ids, k, fixed, frac, iters = jobs[i]
                together[i] = engine.clarans(ids, k, fixed, frac, iters).tolist()
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
        [t.start() for t in threads]
        [t.join() for t in threads]
        assert not errors, errors
        assert together == alone


@pytest.mark.parametrize("tune", ["clarans_draws=40,clarans_slice_us=20,assign_batch_kb=16", "clarans_slice_us=1000000"])
def test_where_a_launch_ends_does_not_change_the_search(tune):
    """LCSGPU_TUNE (read once per process, hence the subprocess): a chain of searches is stopped and started again where it
    stood -- after 40 pre-drawn positions (then twice as many, ...) or 20 microseconds; or with a slice longer than any
    search.  Where a launch ends never changes which step is accepted.  (assign_batch_kb=16: the batched seed assignment
    cut into many launches, jobs split between them.)"""
    import os
    import subprocess
    import sys
    if os.environ.get("LCSGPU_TUNE"):
        pytest.skip("already inside a nested run")
    env = dict(os.environ)
    env["LCSGPU_TUNE"] = tune
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k",
                        "concurrent or family-small or family-indel or two-registers or no-fixed or three-fixed or whole-neighbourhood or ties-everywhere "
                        "or duplicates-tie or tiny or one-medoid or batch"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-3000:]
