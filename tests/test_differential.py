"""Randomised differential tests of the host layer against the REFERENCE ITSELF (oracle/_ref): many small
inputs with heavy distance ties, duplicates, unknown symbols and short sequences -- the regime where tie
breaking and float/double rounding decide the topology.  Needs oracle/_ref (built only where
/root/reference exists), so these run in the build container and are skipped on the GPU box."""
import os

import numpy as np
import pytest

from famsa_amd import hostlib as host_bind
import oracle_bind
from famsa_amd import seqio

pytestmark = pytest.mark.skipif(not oracle_bind.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def host():
    return host_bind.Host()


@pytest.fixture(scope="module")
def ref():
    return oracle_bind.Ref()


def random_set(rng, n, max_len, alphabet, dup_frac):
    seqs = []
    for _ in range(n):
        if seqs and rng.random() < dup_frac:
            s = seqs[int(rng.integers(0, len(seqs)))]
            if rng.random() < 0.5:  # near duplicate
                s = s[: max(1, len(s) - int(rng.integers(0, 3)))]
        else:
            L = int(rng.integers(1, max_len + 1))
            s = "".join(alphabet[i] for i in rng.integers(0, len(alphabet), size=L))
        seqs.append(s)
    # every pair needs a common matching residue (the reference is undefined for LCS 0 in the tree stage)
    seqs = [s + "A" for s in seqs]
    return [f">r{i}" for i in range(n)], seqs


CASES = [(seed, n, max_len, alpha) for seed, (n, max_len, alpha) in enumerate([
    (2, 5, "AC"), (3, 8, "AC"), (7, 12, "ACD"), (16, 20, "AC"), (33, 30, "ARNDX"), (40, 70, "ACDEFGHIKL"),
    (64, 15, "AC"), (65, 40, "ARND"), (90, 100, "ARNDCQEGHILKMFPSTWYVBZX"), (129, 25, "ACD"), (150, 140, "ACDE"),
    (257, 10, "AC")])]


@pytest.mark.parametrize("seed,n,max_len,alpha", CASES)
def test_trees_and_csv_match_reference(host, ref, oracle, tmp_path, seed, n, max_len, alpha):
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    ids, seqs = random_set(rng, n, max_len, alpha, 0.25)
    fasta = str(tmp_path / "in.fasta")
    with open(fasta, "w") as f:
        for i, s in zip(ids, seqs):
            f.write(f"{i}\n{s}\n")
    enc = [oracle.encode(s) for s in seqs]
    codes, offsets = seqio.pack(enc)
    sq = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    h = ref.open_fasta(fasta)
    try:
        for dist_id, dist in [(1, "indel075_div_lcs"), (0, "indel_div_lcs")]:
            for gt in ("sl", "slink", "upgma", "nj", "upgma_modified"):
                for keep in (False, True):
                    want = ref.tree(h, gt, distance=dist_id, keep_dups=int(keep), threads=2)
                    got = host.tree_from_matrix(fasta, sq, gt, distance=dist, keep_duplicates=keep)
                    assert got == want, (gt, dist, keep)
        for square, pid in [(False, False), (True, False), (False, True), (True, True)]:
            a, b = str(tmp_path / "a.csv"), str(tmp_path / "b.csv")
            ref.dist_export(h, a, square=square, pid=pid, threads=2)
            host.dist_export_from_matrix(fasta, sq, b, square_matrix=square, pid=pid)
            assert open(a, "rb").read() == open(b, "rb").read(), (square, pid)
    finally:
        ref.close(h)


@pytest.mark.parametrize("seed,n,max_len,alpha", CASES[:8])
def test_slink_via_mst_matches_reference(host, ref, oracle, tmp_path, monkeypatch, seed, n, max_len, alpha):
    """The MST -> SLINK conversion used on the GPU path, on tie-heavy inputs, against the reference's SLINK."""
    monkeypatch.setenv("FAMSA_HOST_TEST", "slink_from_mst")
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    ids, seqs = random_set(rng, n, max_len, alpha, 0.25)
    fasta = str(tmp_path / "in.fasta")
    with open(fasta, "w") as f:
        for i, s in zip(ids, seqs):
            f.write(f"{i}\n{s}\n")
    enc = [oracle.encode(s) for s in seqs]
    codes, offsets = seqio.pack(enc)
    sq = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    h = ref.open_fasta(fasta)
    try:
        for dist_id, dist in [(1, "indel075_div_lcs"), (0, "indel_div_lcs")]:
            for keep in (False, True):
                want = ref.tree(h, "slink", distance=dist_id, keep_dups=int(keep), threads=2)
                assert host.tree_from_matrix(fasta, sq, "slink", distance=dist, keep_duplicates=keep) == want
    finally:
        ref.close(h)


@pytest.mark.parametrize("seed", range(4))
def test_medoid_and_parttree_match_reference(host, ref, oracle, tmp_path, seed):
    rng = np.random.Generator(np.random.PCG64(77 + seed))
    n = int(rng.integers(150, 400))
    ids, seqs = random_set(rng, n, 60, "ACDEFG", 0.1)
    fasta = str(tmp_path / "in.fasta")
    with open(fasta, "w") as f:
        for i, s in zip(ids, seqs):
            f.write(f"{i}\n{s}\n")
    enc = [oracle.encode(s) for s in seqs]
    codes, offsets = seqio.pack(enc)
    sq = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    h = ref.open_fasta(fasta)
    try:
        for gt in ("sl", "upgma", "nj"):
            for heur_id, heur in [(2, "medoidtree"), (1, "parttree")]:
                kw = dict(subtree=8, sample=40, threshold=30, cluster_fraction=0.3, cluster_iters=2)
                want = ref.tree(h, gt, heuristic=heur_id, threads=3, **kw)
                got = host.tree_from_matrix(fasta, sq, gt, heuristic=heur, subtree_size=8, sample_size=40, threshold=30,
                                            cluster_fraction=0.3, cluster_iters=2)
                assert got == want, (gt, heur)
    finally:
        ref.close(h)


@pytest.mark.parametrize("hooks", ["threads=1", "threads=2,pool=7,leafmax=1", "threads=4,pool=12,leafmax=12", "threads=3,pool=3"])
def test_the_pools_shape_does_not_change_the_tree(host, oracle, tmp_path, monkeypatch, hooks):
    """The FastTree recursion's task pool (host/fasttree.cpp, TaskPool): how many threads it has, how many of them may work
    on leaves while splits wait, or no pool at all -- the tree is assembled by index, so every schedule gives the bytes the
    default gives (which the test above pins to the reference)."""
    rng = np.random.Generator(np.random.PCG64(501))
    n = 600
    ids, seqs = random_set(rng, n, 60, "ACDEFG", 0.1)
    fasta = str(tmp_path / "in.fasta")
    with open(fasta, "w") as f:
        for i, s in zip(ids, seqs):
            f.write(f"{i}\n{s}\n")
    enc = [oracle.encode(s) for s in seqs]
    codes, offsets = seqio.pack(enc)
    sq = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    kw = dict(heuristic="medoidtree", subtree_size=8, sample_size=60, threshold=30, cluster_fraction=0.3, cluster_iters=2)
    want = {gt: host.tree_from_matrix(fasta, sq, gt, **kw) for gt in ("upgma", "sl")}
    monkeypatch.setenv("FAMSA_HOST_TEST", hooks)
    for gt in ("upgma", "sl"):
        assert host.tree_from_matrix(fasta, sq, gt, **kw) == want[gt], gt


@pytest.mark.parametrize("seed", range(3))
def test_num_evals_and_dump_seeds_match_reference(host, ref, oracle, tmp_path, seed):
    """-num_evals 3 (reference core/params.cpp:206: three evaluations per split, the cheapest kept) and -dump_seeds
    (msa.cpp:184-199: the depth-0 seeds' ids) against the reference's own FastTree + observer."""
    rng = np.random.Generator(np.random.PCG64(4100 + seed))
    n = int(rng.integers(180, 420))
    ids, seqs = random_set(rng, n, 60, "ACDEFG", 0.1)
    fasta = str(tmp_path / "in.fasta")
    with open(fasta, "w") as f:
        for i, s in zip(ids, seqs):
            f.write(f"{i}\n{s}\n")
    enc = [oracle.encode(s) for s in seqs]
    codes, offsets = seqio.pack(enc)
    sq = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
    h = ref.open_fasta(fasta)
    try:
        for gt, heur_id, heur, evals in [("upgma", 2, "medoidtree", 3), ("sl", 2, "medoidtree", 2), ("upgma", 1, "parttree", 3),
                                         ("upgma", 2, "medoidtree", 1)]:
            a, b = str(tmp_path / "seeds_ref.txt"), str(tmp_path / "seeds_ours.txt")
            want = ref.tree(h, gt, heuristic=heur_id, threads=3, subtree=8, sample=40, threshold=30, cluster_fraction=0.3,
                            cluster_iters=2, num_evals=evals, dump_seeds=a)
            got = host.tree_from_matrix(fasta, sq, gt, heuristic=heur, subtree_size=8, sample_size=40, threshold=30,
                                        cluster_fraction=0.3, cluster_iters=2, num_evals=evals, dump_seeds=b)
            assert got == want, (gt, heur, evals)
            assert open(a).read() == open(b).read() and len(open(a).read().split()) == 8, (gt, heur, evals)
    finally:
        ref.close(h)


def _random_triangle(rng, n, levels):
    """Float distance triangle; `levels` > 0 quantises the values so that ties are everywhere."""
    m = n * (n - 1) // 2
    d = rng.random(m, dtype=np.float32) * np.float32(3.0) + np.float32(0.01)
    if levels:
        d = (np.floor(d * levels) / np.float32(levels)).astype(np.float32) + np.float32(1.0 / levels)
    return d


@pytest.mark.parametrize("seed,n,k,fixed,frac,iters,levels", [
    (0, 40, 5, 1, 0.1, 2, 0), (1, 40, 5, 0, 0.5, 3, 4), (2, 120, 12, 1, 0.1, 2, 0), (3, 120, 12, 3, 1.0, 1, 8),
    (4, 300, 30, 1, 0.1, 2, 0), (5, 300, 30, 1, 0.1, 2, 16), (6, 12, 12, 1, 0.1, 2, 0), (7, 9, 1, 0, 0.1, 2, 0),
    (8, 600, 100, 1, 0.05, 2, 0), (9, 64, 2, 1, 0.3, 4, 2),
])
def test_host_clarans_matches_reference(host, ref, seed, n, k, fixed, frac, iters, levels):
    """The host CLARANS search against the reference's CLARANS::operator() on the same float triangle."""
    rng = np.random.default_rng(9100 + seed)
    tri = _random_triangle(rng, n, levels)
    want = ref.clarans(tri, n, k, fixed, frac, iters)
    got = host.clarans(tri, n, k, fixed, frac, iters)
    assert got.tolist() == want.tolist()
