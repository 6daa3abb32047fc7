"""CPU tests of the C++ host layer (famsa_amd/host): guide-tree builders, duplicate handling, Newick and
CSV writers -- fed with LCS matrices from the ORACLE (no GPU needed), compared byte for byte with the
reference's own golden files and with goldens generated from the reference (oracle/make_golden.py)."""
import os

import numpy as np
import pytest

from famsa_amd import hostlib as host_bind
import oracle_bind
from famsa_amd import seqio

G = oracle_bind.GOLDEN


@pytest.fixture(scope="module")
def host():
    return host_bind.Host()


_cache = {}


def square(oracle, fasta, symmetric_ok=True):
    """Full oriented square LCS matrix in input order from the oracle."""
    if fasta in _cache:
        return _cache[fasta]
    ids, seqs = seqio.read_fasta(fasta)
    enc = [oracle.encode(s) for s in seqs]
    codes, offsets = seqio.pack(enc)
    n = len(enc)
    if symmetric_ok and n > 1000:
        # no carry-quirk sequences in these sets: one triangle is enough (checked below on a sample)
        tri = oracle.triangle(codes, offsets)
        m = np.zeros((n, n), np.uint32)
        il = np.tril_indices(n, -1)
        m[il] = tri
        m = m + m.T
        m[np.arange(n), np.arange(n)] = [len(e) if (e < 20).all() else int((e < 20).sum()) for e in enc]
        sample = np.arange(0, n, 211)
        assert (oracle.rect(codes, offsets, sample, sample) == m[np.ix_(sample, sample)]).all()
    else:
        m = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    _cache[fasta] = m
    return m


def test_number_format(host, oracle):
    for v in [0.0, 0.674734, 1.0, 0.9999995, 12.5, 123456.789, 3.4e38, float(np.float32(1e30)), 0.0000004]:
        assert host.format_distance(v) == oracle.format_dist(v)


@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "nj"])
def test_adeno_trees_vs_reference_goldens(host, oracle, gt):
    f = os.path.join(G, "adeno_fiber", "adeno_fiber")
    got = host.tree_from_matrix(f, square(oracle, f), gt)
    assert got == open(os.path.join(G, "adeno_fiber", gt + ".dnd"), "rb").read()


def test_adeno_duplicates_tree(host, oracle):
    f = os.path.join(G, "adeno_fiber_duplicates", "adeno_fiber_duplicates")
    got = host.tree_from_matrix(f, square(oracle, f), "sl")
    assert got == open(os.path.join(G, "adeno_fiber_duplicates", "sl.dnd"), "rb").read()


@pytest.mark.parametrize("name,sq,pid", [("dist", False, False), ("pid", False, True), ("dist_sq", True, False),
                                         ("pid_sq", True, True)])
def test_adeno_dist_export_vs_reference_goldens(host, oracle, tmp_path, name, sq, pid):
    f = os.path.join(G, "adeno_fiber", "adeno_fiber")
    out = str(tmp_path / "o.csv")
    host.dist_export_from_matrix(f, square(oracle, f), out, square_matrix=sq, pid=pid)
    assert open(out, "rb").read() == open(os.path.join(G, "adeno_fiber", name + ".csv"), "rb").read()


@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "nj"])
def test_adversarial_trees_with_carry_quirk(host, oracle, gt):
    """Orientation-sensitive refs: MSTPrim needs LCS(ref = node just added, partner = candidate)."""
    f = os.path.join(G, "adversarial_tree.fasta")
    m = square(oracle, f, symmetric_ok=False)
    assert (m != m.T).any()
    got = host.tree_from_matrix(f, m, gt)
    assert got == open(os.path.join(G, f"adversarial_tree_{gt}.dnd"), "rb").read()


@pytest.mark.parametrize("name,gold", [("adeno_fiber/adeno_fiber", "adeno_fiber/sl.dnd"),
                                       ("adversarial_tree.fasta", "adversarial_tree_sl.dnd")])
def test_prim_with_one_row_per_step(host, oracle, monkeypatch, name, gold):
    """-gt sl when the triangle is not held anywhere: every Prim step asks the source for the row of the node just
    added against the unprocessed vertices (MSTPrim::run_view's own shape, reference tree/MSTPrim.cpp:356-533) --
    O(n) memory; the orientation-sensitive set shows that ref = the node just added is honoured."""
    monkeypatch.setenv("FAMSA_HOST_TEST", "prim_streaming")
    f = os.path.join(G, name)
    m = square(oracle, f, symmetric_ok=(name != "adversarial_tree.fasta"))
    assert host.tree_from_matrix(f, m, "sl") == open(os.path.join(G, gold), "rb").read()


@pytest.mark.parametrize("pid", [False, True])
def test_dist_export_row_blocks_and_column_order(host, oracle, tmp_path, monkeypatch, pid):
    """-dist_export asks for its rows in blocks, each as a rectangle whose columns are in LENGTH order, and writes the text
    column by original column (host/trees.cpp, write_csv_d).  700 ragged sequences in no particular order = three row blocks
    of the triangle form: every row of the triangle must be the head of the same row of the square form (one block; that form
    with permuted columns is pinned to the reference's files by the tests above), and the export with the columns as they were
    read must be the same bytes."""
    rng = np.random.default_rng(77)
    lens = rng.integers(5, 60, size=700)
    fasta = str(tmp_path / "ragged.fasta")
    with open(fasta, "w") as f:
        for i, n in enumerate(lens):
            f.write(f">r{i}\n" + "".join("ACDEFGHIKLMNPQRSTVWY"[c] for c in rng.integers(0, 20, size=int(n))) + "\n")
    m = square(oracle, fasta, symmetric_ok=False)
    tri, sq, tri_in = (str(tmp_path / n) for n in ("tri.csv", "sq.csv", "tri_input.csv"))
    host.dist_export_from_matrix(fasta, m, tri, pid=pid)
    host.dist_export_from_matrix(fasta, m, sq, square_matrix=True, pid=pid)
    monkeypatch.setenv("FAMSA_HOST_TEST", "csv_input_order")
    host.dist_export_from_matrix(fasta, m, tri_in, pid=pid)
    assert open(tri, "rb").read() == open(tri_in, "rb").read()
    rows_tri = open(tri).read().split("\n")[:-1]
    rows_sq = open(sq).read().split("\n")[1:-1]  # (the square form starts with a header line)
    assert len(rows_tri) == len(rows_sq) == 700
    for i, (a, b) in enumerate(zip(rows_tri, rows_sq)):
        assert a.split(",") == b.split(",")[: i + 1], i


def test_adversarial_csv_zero_lcs(host, oracle, tmp_path):
    f = os.path.join(G, "adversarial.fasta")
    m = square(oracle, f, symmetric_ok=False)
    out = str(tmp_path / "o.csv")
    host.dist_export_from_matrix(f, m, out, square_matrix=True)
    assert open(out, "rb").read() == open(os.path.join(G, "adversarial_dist_sq.csv"), "rb").read()
    host.dist_export_from_matrix(f, m, out, pid=True)
    assert open(out, "rb").read() == open(os.path.join(G, "adversarial_pid.csv"), "rb").read()


@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "nj"])
def test_hemopexin_trees(host, oracle, gt):
    f = os.path.join(G, "hemopexin", "hemopexin")
    got = host.tree_from_matrix(f, square(oracle, f), gt)
    assert got == open(os.path.join(G, "hemopexin", gt + ".dnd"), "rb").read()


@pytest.mark.parametrize("name", ["one-seq", "two-seq", "many-seq"])
def test_dummy_inputs(host, oracle, name):
    """1 / 2 / many identical sequences (reference .github/workflows/self-hosted.yml:126-149)."""
    f = os.path.join(G, "dummy", name)
    ids, seqs = seqio.read_fasta(f)
    got = host.tree_from_matrix(f, square(oracle, f), "sl")
    # all records of these files are identical: after duplicate removal one unique sequence is
    # left and the reference leaves the tree stage without a tree (msa.cpp:549-556)
    assert got == b""
    got_keep = host.tree_from_matrix(f, square(oracle, f), "sl", keep_duplicates=True)
    if len(ids) > 1:
        for i in ids:
            assert i[1:].encode() in got_keep


def test_workset_matches_python_order(host, oracle):
    f = os.path.join(G, "adeno_fiber_duplicates", "adeno_fiber_duplicates")
    ids, seqs = seqio.read_fasta(f)
    enc = [oracle.encode(s) for s in seqs]
    u, s2i, s2u = host.workset(f, len(ids))
    assert list(s2i) == seqio.sort_order(enc)
    assert u == len({bytes(e) for e in enc})


# ---- MedoidTree / PartTree heuristic (reference tree/FastTree.cpp, tree/Clustering.cpp) ----------------
@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "nj"])
def test_hemopexin_medoid_trees_vs_reference_goldens(host, oracle, gt):
    """The reference's own goldens test/hemopexin/medoid-*.dnd (CLARANS seeds, assignment, stitching)."""
    f = os.path.join(G, "hemopexin", "hemopexin")
    got = host.tree_from_matrix(f, square(oracle, f), gt, heuristic="medoidtree")
    assert got == open(os.path.join(G, "hemopexin", f"medoid-{gt}.dnd"), "rb").read()


@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "nj"])
def test_hemopexin_medoid_nondefault_params(host, oracle, gt):
    """.github/workflows/main.yml:136-139: -subtree_size 10 -sample_size 100 -medoid_threshold 100
    -cluster_fraction 0.2 -cluster_iters 1 (deep recursion, sampled clustering)."""
    f = os.path.join(G, "hemopexin", "hemopexin")
    got = host.tree_from_matrix(f, square(oracle, f), gt, heuristic="medoidtree", subtree_size=10, sample_size=100,
                                threshold=100, cluster_fraction=0.2, cluster_iters=1)
    assert got == open(os.path.join(G, "hemopexin", f"medoid-{gt}-params.dnd"), "rb").read()


@pytest.mark.parametrize("keep,name", [(False, "medoid-sl.dnd"), (True, "medoid-sl-dups.dnd")])
def test_hemopexin_duplicates_medoid(host, oracle, keep, name):
    f = os.path.join(G, "hemopexin_duplicates", "hemopexin_duplicates")
    got = host.tree_from_matrix(f, square(oracle, f), "sl", heuristic="medoidtree", keep_duplicates=keep)
    assert got == open(os.path.join(G, "hemopexin_duplicates", name), "rb").read()


@pytest.mark.skipif(not oracle_bind.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_parttree_vs_reference_library(host, oracle):
    """-parttree has no upstream golden: compare with the reference itself (random seeds path)."""
    f = os.path.join(G, "hemopexin", "hemopexin")
    ref = oracle_bind.Ref()
    h = ref.open_fasta(f)
    want = ref.tree(h, "upgma", heuristic=1, threshold=100)
    ref.close(h)
    got = host.tree_from_matrix(f, square(oracle, f), "upgma", heuristic="parttree", threshold=100)
    assert got == want


def test_medoid_subtrees_built_by_worker_threads(host, oracle, monkeypatch):
    """The top-level sub-trees run on worker threads (as in the reference); the result must not change."""
    monkeypatch.setenv("FAMSA_HOST_TEST", "threads=6")
    f = os.path.join(G, "hemopexin", "hemopexin")
    got = host.tree_from_matrix(f, square(oracle, f), "upgma", heuristic="medoidtree", subtree_size=10,
                                sample_size=100, threshold=100, cluster_fraction=0.2, cluster_iters=1)
    assert got == open(os.path.join(G, "hemopexin", "medoid-upgma-params.dnd"), "rb").read()


def test_slink_equals_canonical_pointer_representation_of_the_mst(host, oracle, monkeypatch):
    """On the GPU `-gt slink` = device Prim (triangle-orientation distances) + an O(n log n) conversion:
    SLINK's output is the canonical pointer representation of the single-linkage hierarchy under the
    strict (distance, packed ids) order.  Here the same conversion runs with Prim on the host."""
    monkeypatch.setenv("FAMSA_HOST_TEST", "slink_from_mst")
    for name in ("adeno_fiber/adeno_fiber", "adversarial_tree.fasta", "hemopexin/hemopexin"):
        f = os.path.join(G, name)
        gold = {"adeno_fiber/adeno_fiber": "adeno_fiber/slink.dnd", "adversarial_tree.fasta": "adversarial_tree_slink.dnd",
                "hemopexin/hemopexin": "hemopexin/slink.dnd"}[name]
        m = square(oracle, f, symmetric_ok=(name != "adversarial_tree.fasta"))
        assert host.tree_from_matrix(f, m, "slink") == open(os.path.join(G, gold), "rb").read()


def _reader_restatement(raw):
    """The reference's loadFasta state machine (core/io_service.h:99-124), line by line."""
    ids, seqs, ident, seq = [], [], "", ""
    for line in raw.split(b"\n"):
        line = line.decode("latin-1").rstrip("\r\n")
        if not line:
            continue
        if line[0] == ">":
            if ident and seq:
                ids.append(ident)
                seqs.append(seq)
                seq = ""
            ident = line
        else:
            seq += line
    if ident and seq:
        ids.append(ident)
        seqs.append(seq)
    return ids, seqs


@pytest.mark.parametrize("variant", ["plain", "junk-first", "crlf", "no-final-newline", "only-junk"])
def test_parallel_fasta_reader_matches_the_serial_semantics(host, oracle, tmp_path, variant):
    """The reader splits the file at header lines and parses the pieces on several threads: same
    records as the serial state machine (and as the reference's own loader), with headers that have
    no residues, blank lines, multi-line and lower-case records, and residues before any header."""
    rng = np.random.Generator(np.random.PCG64(41))
    alpha = np.frombuffer(b"ARNDCQEGHILKMFPSTWYVBZX*acdxy-", dtype=np.uint8)
    eol = b"\r\n" if variant == "crlf" else b"\n"
    chunks = []
    if variant in ("junk-first", "only-junk"):
        chunks.append(b"MKV" + eol + b"LLA" + eol)
    n = 0 if variant == "only-junk" else 9000
    for i in range(n):
        chunks.append(b">seq%d some description" % i + eol)
        if i % 97 == 5:
            continue  # a header without residues: the next header replaces it
        if i % 101 == 7:
            chunks.append(b"--" + eol + b"-" + eol)  # residue lines of gaps only: kept, as a sequence of length 0
            continue
        length = int(rng.integers(1, 700))
        res = alpha[rng.integers(0, len(alpha), size=length)].tobytes()
        for a in range(0, length, 60):
            chunks.append(res[a:a + 60] + eol)
            if rng.random() < 0.01:
                chunks.append(eol)
    raw = b"".join(chunks)
    if variant == "no-final-newline":
        raw = raw.rstrip(b"\r\n")
    path = str(tmp_path / "reader.fasta")
    with open(path, "wb") as f:
        f.write(raw)
    want_ids, want_seqs = _reader_restatement(raw)
    assert variant == "only-junk" or len(raw) > 3 << 20  # several pieces
    for threads in (1, 7):
        ids, codes = host.records(path, threads)
        assert ids == want_ids
        assert len(codes) == len(want_seqs)
        for got, residues in zip(codes[:400] + codes[-400:], want_seqs[:400] + want_seqs[-400:]):
            assert (got == oracle.encode(residues)).all()
    if oracle_bind.have_ref() and variant != "only-junk":
        ref = oracle_bind.Ref()
        h = ref.open_fasta(path)
        rc = ref.codes(h)
        ref.close(h)
        assert len(rc) == len(codes)
        assert all((a == b).all() for a, b in zip(rc, codes))


def test_gzip_input_is_read_like_plain_text(host, tmp_path):
    """A .gz input (one member, or several concatenated members as bgzip writes them) gives the same records."""
    import gzip
    rng = np.random.Generator(np.random.PCG64(43))
    alpha = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
    parts = []
    for i in range(6000):
        parts.append(b">g%d\n" % i + alpha[rng.integers(0, 20, size=int(rng.integers(5, 600)))].tobytes() + b"\n")
    raw = b"".join(parts)
    plain, one, many = (str(tmp_path / n) for n in ("p.fasta", "one.fasta.gz", "many.fasta.gz"))
    open(plain, "wb").write(raw)
    open(one, "wb").write(gzip.compress(raw))
    cut = [0, len(raw) // 3, 2 * len(raw) // 3 + 11, len(raw)]  # member borders inside records
    open(many, "wb").write(b"".join(gzip.compress(raw[a:b]) for a, b in zip(cut, cut[1:])))
    want_ids, want_codes = host.records(plain, 3)
    for path in (one, many):
        ids, codes = host.records(path, 3)
        assert ids == want_ids
        assert all((a == b).all() for a, b in zip(codes, want_codes)) and len(codes) == len(want_codes)
    bad = str(tmp_path / "bad.fasta.gz")
    open(bad, "wb").write(gzip.compress(raw)[: 5000])
    with pytest.raises(RuntimeError):
        host.records(bad, 2)


@pytest.mark.parametrize("n,spread,zeros", [(130, 2, False), (257, 3, True), (700, 3, False), (1300, 40, True)])
def test_the_two_host_upgma_forms_build_the_same_tree(host, tmp_path, monkeypatch, n, spread, zeros):
    """UPGMA over a host-side matrix has two forms (trees.cpp): the triangle walk and, up to 4096 rows -- every leaf
    of the FastTree recursion --, a square float matrix with SSE sweeps, a lazy heap and packing when half of the
    rows are gone.  Same tree on tie-heavy random 'LCS' values (few distinct distances: every tie rule is hit), with
    pairs that share nothing (distance FLT_MAX, never picked while a finite one is left) and across the packing
    thresholds.  (Both forms against the reference itself: test_differential.py, n <= 257.)"""
    rng = np.random.Generator(np.random.PCG64(77 + n))
    lens = np.sort(rng.integers(250, 301, size=n))[::-1]
    fasta = str(tmp_path / "in.fasta")
    with open(fasta, "w") as f:
        for i, L in enumerate(lens):
            f.write(f">s{i}\n" + "".join("ACDEFGHIKLMNPQRSTVWY"[c] for c in rng.integers(0, 20, size=int(L))) + "\n")
    m = rng.integers(90, 90 + spread, size=(n, n), dtype=np.uint32)
    m = np.tril(m, -1)
    if zeros:  # a few sequences share nothing with a few others
        for a, b in rng.integers(1, n, size=(n // 20, 2)):
            if a != b:
                m[max(a, b), min(a, b)] = 0
    m = np.ascontiguousarray(m + m.T)
    m[np.arange(n), np.arange(n)] = lens
    for gt in ("upgma", "upgma_modified"):
        for dist in ("indel075_div_lcs", "indel_div_lcs"):
            monkeypatch.delenv("FAMSA_HOST_TEST", raising=False)
            square_form = host.tree_from_matrix(fasta, m, gt, distance=dist, keep_duplicates=True)
            monkeypatch.setenv("FAMSA_HOST_TEST", "upgma_triangle")
            triangle_form = host.tree_from_matrix(fasta, m, gt, distance=dist, keep_duplicates=True)
            assert square_form == triangle_form, (gt, dist)
            assert square_form.count(b",") == n - 1


def _newick_walk(left, right, names):
    """the recursive definition (reference tree/NewickParser.cpp:103-165), iteratively"""
    n = len(names)
    text, stack = [], [(n + len(left) - 1, 0)]
    root = stack[0][0]
    while stack:
        node, state = stack.pop()
        if node < n:
            text.append(names[node].lstrip(">") if names[node].startswith(">") else names[node])
            text.append(":1.0")
        elif state == 0:
            text.append("(")
            stack.append((node, 1))
            stack.append((int(left[node - n]), 0))
        elif state == 1:
            text.append(",")
            stack.append((node, 2))
            stack.append((int(right[node - n]), 0))
        else:
            text.append(");" if node == root else "):1.0")
    return "".join(text).encode()


@pytest.mark.parametrize("n,shape", [(2, "random"), (3, "chain"), (1000, "random"), (150000, "random"), (150000, "chain")])
def test_newick_writer_closed_form_equals_the_walk(host, n, shape):
    """The writer places every node's characters from lengths and positions computed in two sweeps and fills the text
    on all cores (trees.cpp, tree_to_newick); the text is the recursive definition's, for bushy trees and for a chain
    of 150 000 (no recursion anywhere), with names of different lengths and a leading '>' dropped."""
    rng = np.random.Generator(np.random.PCG64(n))
    names = [(">" if i % 3 == 0 else "") + "s%d" % (i * 7919 % 100003) + "x" * (i % 5) for i in range(n)]
    left, right = np.zeros(n - 1, np.int32), np.zeros(n - 1, np.int32)
    roots = list(range(n))
    for k in range(n - 1):
        if shape == "chain":
            a, b = (roots.pop(), roots.pop()) if k == 0 else (roots.pop(0), roots.pop())
        else:
            a = roots.pop(int(rng.integers(0, len(roots))))
            b = roots.pop(int(rng.integers(0, len(roots))))
        left[k], right[k] = a, b
        roots.append(n + k)
    assert host.newick(left, right, names) == _newick_walk(left, right, [s.lstrip(">") for s in names])


def test_newick_writer_takes_any_node_order(host):
    """a child stored after its parent (no generator of ours does that): the writer falls back to the walk"""
    names = ["a", "b", "c", "d"]
    # nodes 4 = (5, 3), 5 = (0, 1), 6 = root = (4, 2): node 4 comes before its child 5
    left, right = np.array([5, 0, 4], np.int32), np.array([3, 1, 2], np.int32)
    assert host.newick(left, right, names) == b"(((a:1.0,b:1.0):1.0,d:1.0):1.0,c:1.0);"
    # and the same tree in the usual order: 4 = (0, 1), 5 = (4, 3), 6 = (5, 2)
    assert host.newick([0, 4, 5], [1, 3, 2], names) == b"(((a:1.0,b:1.0):1.0,d:1.0):1.0,c:1.0);"


def test_duplicates_are_reattached_like_the_reference(host):
    """GuideTree::fromUnique over a map with runs of duplicates (the CSR form of trees.cpp against the definition:
    the first two records of a run become a node, every further one joins (record, previous node))"""
    # records 0..6 in sorted order; unique sequences: u0 = {0}, u1 = {1, 2, 3}, u2 = {4}, u3 = {5, 6}
    s2u = [0, 1, 1, 1, 2, 3, 3]
    names = ["r%d" % i for i in range(7)]
    # unique tree: leaves 0..3, internal 4 = (0, 1), 5 = (2, 3), 6 = (4, 5)
    left, right = [0, 2, 4], [1, 3, 5]
    got = host.newick(left, right, names, sorted2unique=s2u)
    u1 = "(r3:1.0,(r1:1.0,r2:1.0):1.0):1.0"
    u3 = "(r5:1.0,r6:1.0):1.0"
    assert got == ("((r0:1.0,%s):1.0,(r4:1.0,%s):1.0);" % (u1, u3)).encode()


def test_working_order_on_a_set_that_runs_the_parallel_sort(host, oracle, tmp_path):
    """famsa_order sorts 16-byte keys (length, the first 12 residue codes, index) in parallel runs and merges them; the
    sequences are looked at only where length and prefix tie.  20 000 records with few lengths, shared 12-residue
    prefixes, prefixes that differ in the last key residue, records shorter than the prefix, and exact duplicates:
    the order is the reference's stable sort (length descending, codes ascending), the duplicate map the adjacent-equal runs."""
    rng = np.random.Generator(np.random.PCG64(4242))
    alphabet = "ARNDCQEGHILKMFPSTWYVBZX"
    stems = ["".join(alphabet[c] for c in rng.integers(0, 23, size=12)) for _ in range(6)]
    seqs = []
    for i in range(20000):
        kind = i % 5
        L = int(rng.choice([5, 11, 12, 13, 40, 41]))
        if kind == 0 and seqs:
            s = seqs[int(rng.integers(0, len(seqs)))]  # exact duplicate
        elif kind == 1:
            stem = stems[int(rng.integers(0, 6))]
            s = (stem + "".join(alphabet[c] for c in rng.integers(0, 3, size=40)))[:L]  # tie on the prefix
        elif kind == 2:
            stem = stems[int(rng.integers(0, 6))]
            s = (stem[:11] + alphabet[int(rng.integers(0, 23))] + "ACAC" * 10)[:L]  # differ at the key's last residue
        else:
            s = "".join(alphabet[c] for c in rng.integers(0, 23, size=L))
        seqs.append(s)
    f = str(tmp_path / "order.fasta")
    with open(f, "w") as out:
        for i, s in enumerate(seqs):
            out.write(">q%d\n%s\n" % (i, s))
    enc = [oracle.encode(s) for s in seqs]
    want = seqio.sort_order(enc)
    u, s2i, s2u = host.workset(f, len(seqs))
    assert list(s2i) == want
    assert u == len({bytes(e) for e in enc})
    expect_u, cur, prev = [], -1, None
    for k in want:
        if prev is None or bytes(enc[k]) != prev:
            cur += 1
            prev = bytes(enc[k])
        expect_u.append(cur)
    assert list(s2u) == expect_u
    u_keep, s2i_keep, s2u_keep = host.workset(f, len(seqs), keep_duplicates=True)
    assert list(s2i_keep) == want and u_keep == len(seqs) and list(s2u_keep) == list(range(len(seqs)))


def _mt19937(seed):
    """The mt19937 stream (32-bit outputs) -- numpy's bit generator seeded the std::mt19937 way."""
    from numpy.random import MT19937
    bg = MT19937()
    st = bg.state
    key = np.zeros(624, np.uint32)
    key[0] = seed & 0xFFFFFFFF
    for i in range(1, 624):
        key[i] = (1812433253 * (int(key[i - 1]) ^ (int(key[i - 1]) >> 30)) + i) & 0xFFFFFFFF
    st["state"]["key"] = key
    st["state"]["pos"] = 624
    bg.state = st
    return lambda: int(bg.random_raw())


@pytest.mark.parametrize("seed", [0, 1, 77])
def test_chained_tree_is_a_caterpillar_over_the_seeded_order(host, oracle, seed):
    """-gt chained [seed] (reference tree/Chained.h:8-35, developer builds): node n = (idx[0], idx[1]), every later node
    (idx[i], previous node); the order here is a Fisher-Yates shuffle by mt19937(seed) through the reference's
    det_uniform_int_distribution -- restated below -- over the sorted unique working order; no distance is looked at."""
    f = os.path.join(G, "adeno_fiber", "adeno_fiber")
    ids, seqs = seqio.read_fasta(f)
    n = len(ids)
    zeros = np.zeros((n, n), np.uint32)  # the matrix must not matter
    got = host.tree_from_matrix(f, zeros, "chained", chained_seed=seed)
    u, s2i, s2u = host.workset(f, n)
    assert u == n  # no duplicates in this set: leaf k of the tree = record s2i[k]
    g = _mt19937(seed)
    idx = list(range(n))
    for i in range(n - 1):
        diff = n - i
        bad = 0xFFFFFFFF // diff
        while True:
            r = g()
            if r // diff < bad:
                break
        j = i + r % diff
        idx[i], idx[j] = idx[j], idx[i]
    name = lambda k: ids[s2i[k]].lstrip(">")
    text = "(%s:1.0,%s:1.0)" % (name(idx[0]), name(idx[1]))
    for i in range(2, n):
        text = "(%s:1.0,%s:1.0)" % (name(idx[i]), text)
    assert got.decode() == text + ";"
    assert host.tree_from_matrix(f, zeros, "chained", chained_seed=seed + 1) != got
    with pytest.raises(RuntimeError, match="Illegal guide tree method"):  # nothing to wrap (reference msa.cpp:170)
        host.tree_from_matrix(f, zeros, "chained", heuristic="medoidtree", threshold=10)


def test_newick_entry_point_refuses_malformed_trees(host):
    """famsa_host_newick is exported: a child out of range, a node with two parents, the wrong number of names are errors,
    not out-of-bounds reads."""
    names = ["a", "b", "c"]
    assert host.newick([0, 3], [1, 2], names) == b"((a:1.0,b:1.0):1.0,c:1.0);"
    for left, right, nm in [([0, 3], [1, 7], names), ([0, 0], [1, 2], names), ([0, -1], [1, 2], names), ([0, 3], [1, 2], names[:2]),
                            ([0, 4], [1, 2], names)]:
        with pytest.raises(RuntimeError, match="famsa_host_newick"):
            host.newick(left, right, nm)
