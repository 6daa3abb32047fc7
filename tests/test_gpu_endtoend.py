"""GPU end-to-end tests: FASTA in -> Newick / CSV out through the C++ host layer and the famsa-gpu CLI,
LCS computed by the HIP kernels; byte-identical to the reference's outputs (committed goldens)."""
import hashlib
import json
import os
import subprocess

import pytest

from famsa_amd import hostlib as host_bind
import oracle_bind
from famsa_amd import seqio

pytestmark = pytest.mark.gpu
G = oracle_bind.GOLDEN


@pytest.fixture(scope="module")
def host():
    return host_bind.Host()


def run_cli(*args):
    p = subprocess.run([host_bind.CLI, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr
    return p


@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "nj"])
def test_cli_adeno_tree(tmp_path, gt):
    out = str(tmp_path / "t.dnd")
    run_cli("-gt", gt, "-gt_export", os.path.join(G, "adeno_fiber", "adeno_fiber"), out)
    assert open(out, "rb").read() == open(os.path.join(G, "adeno_fiber", gt + ".dnd"), "rb").read()


def test_cli_default_is_sl_and_verbose_stats(tmp_path):
    out = str(tmp_path / "t.dnd")
    p = run_cli("-v", "-gt_export", os.path.join(G, "adeno_fiber", "adeno_fiber"), out)
    assert open(out, "rb").read() == open(os.path.join(G, "adeno_fiber", "sl.dnd"), "rb").read()
    assert "time.tree_build=" in p.stderr and "gpu.lcs_kernel_ms=" in p.stderr


@pytest.mark.skipif(not oracle_bind.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_cli_flags_added_in_round_5(tmp_path):
    """-num_evals, -dump_seeds, -keep_duplicates, -stats, -shuffle and -gt chained on the command line (reference
    core/params.cpp:178-254): hemopexin's MedoidTree with three evaluations per split and its depth-0 seeds against the
    reference's own FastTree + observer; the statistics file's shape; a chained tree per seed."""
    f = os.path.join(G, "hemopexin", "hemopexin")
    ref = oracle_bind.Ref()
    h = ref.open_fasta(f)
    try:
        for gt, evals, keep in [("upgma", 3, False), ("sl", 2, True)]:
            want_seeds = str(tmp_path / "seeds_ref.txt")
            want = ref.tree(h, gt, heuristic=2, threads=8, num_evals=evals, dump_seeds=want_seeds, keep_dups=int(keep))
            out, seeds, stats = str(tmp_path / "m.dnd"), str(tmp_path / "seeds.txt"), str(tmp_path / "stats.txt")
            run_cli("-medoidtree", "-gt", gt, "-num_evals", str(evals), "-dump_seeds", seeds, "-stats", stats, "-shuffle", "7",
                    *(["-keep_duplicates"] if keep else []), "-gt_export", f, out)
            assert open(out, "rb").read() == want
            assert open(seeds).read() == open(want_seeds).read() and len(open(seeds).read().split()) == 100
            lines = open(stats).read().splitlines()
            assert lines[0] == "[stats]" and lines[1:] == sorted(lines[1:]) and all("=" in ln for ln in lines[1:])
            keys = {ln.split("=")[0] for ln in lines[1:]}
            assert {"input.n_sequences", "input.n_duplicates", "time.sort", "time.tree_build", "time.tree_store", "time.total"} <= keys
            assert "input.n_sequences=4188" in lines
    finally:
        ref.close(h)
    a, b, c = (str(tmp_path / n) for n in ("c0.dnd", "c1.dnd", "c0again.dnd"))
    run_cli("-gt", "chained", "-gt_export", f, a)
    run_cli("-gt", "chained", "5", "-gt_export", f, b)
    run_cli("-gt", "chained", "0", "-gt_export", f, c)
    ta, tb = open(a).read(), open(b).read()
    assert ta == open(c).read() and ta != tb and ta.count(",") == 4187 and ta.endswith(");")
    p = subprocess.run([host_bind.CLI, "-gt", "chained", "-medoidtree", "-gt_export", f, a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "Illegal guide tree method" in p.stderr


@pytest.mark.parametrize("name,flags", [("dist", []), ("pid", ["-pid"]), ("dist_sq", ["-square_matrix"]),
                                        ("pid_sq", ["-square_matrix", "-pid"])])
def test_cli_adeno_dist_export(tmp_path, name, flags):
    out = str(tmp_path / "d.csv")
    run_cli("-dist_export", *flags, os.path.join(G, "adeno_fiber", "adeno_fiber"), out)
    assert open(out, "rb").read() == open(os.path.join(G, "adeno_fiber", name + ".csv"), "rb").read()


def test_cli_duplicates(tmp_path):
    out = str(tmp_path / "t.dnd")
    run_cli("-gt", "sl", "-gt_export", os.path.join(G, "adeno_fiber_duplicates", "adeno_fiber_duplicates"), out)
    assert open(out, "rb").read() == open(os.path.join(G, "adeno_fiber_duplicates", "sl.dnd"), "rb").read()


@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "nj"])
def test_hemopexin_tree(host, gt):
    got = host.tree_gpu(os.path.join(G, "hemopexin", "hemopexin"), gt)
    assert got == open(os.path.join(G, "hemopexin", gt + ".dnd"), "rb").read()


def test_hemopexin_dist_export_checksum(host, tmp_path):
    """BASELINE config 2: hemopexin -dist_export, 79 MB of CSV, byte-identical (sha256 from the reference)."""
    meta = json.load(open(os.path.join(G, "meta.json")))["hemopexin"]
    out = str(tmp_path / "d.csv")
    host.dist_export_gpu(os.path.join(G, "hemopexin", "hemopexin"), out)
    data = open(out, "rb").read()
    assert len(data) == meta["dist_csv_bytes"]
    assert hashlib.sha256(data).hexdigest() == meta["dist_csv_sha256"]


@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "nj"])
def test_adversarial_tree_with_carry_quirk(host, gt):
    got = host.tree_gpu(os.path.join(G, "adversarial_tree.fasta"), gt)
    assert got == open(os.path.join(G, f"adversarial_tree_{gt}.dnd"), "rb").read()


def test_adversarial_csv(host, tmp_path):
    out = str(tmp_path / "d.csv")
    f = os.path.join(G, "adversarial.fasta")
    host.dist_export_gpu(f, out, square_matrix=True)
    assert open(out, "rb").read() == open(os.path.join(G, "adversarial_dist_sq.csv"), "rb").read()
    host.dist_export_gpu(f, out, pid=True)
    assert open(out, "rb").read() == open(os.path.join(G, "adversarial_pid.csv"), "rb").read()


@pytest.mark.parametrize("gt", ["sl", "upgma"])
def test_synthetic_2k_tree(host, tmp_path, gt):
    codes, offsets = seqio.synth_uniform(2000, 400)
    f = str(tmp_path / "s.fasta")
    seqio.to_fasta(codes, offsets, f)
    assert host.tree_gpu(f, gt) == open(os.path.join(G, f"synth2k_{gt}.dnd"), "rb").read()


def test_cli_refuses_out_of_scope_flags(tmp_path):
    p = subprocess.run([host_bind.CLI, "-gz", "-gt_export", os.path.join(G, "adeno_fiber", "adeno_fiber"),
                        str(tmp_path / "x")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "outside the scope" in p.stderr


@pytest.mark.parametrize("gt", ["sl", "upgma", "nj"])
def test_hemopexin_medoidtree(host, gt):
    got = host.tree_gpu(os.path.join(G, "hemopexin", "hemopexin"), gt, heuristic="medoidtree")
    assert got == open(os.path.join(G, "hemopexin", f"medoid-{gt}.dnd"), "rb").read()


def test_cli_medoidtree_nondefault_params(tmp_path):
    out = str(tmp_path / "t.dnd")
    run_cli("-medoidtree", "-gt", "slink", "-gt_export", "-subtree_size", "10", "-sample_size", "100",
            "-medoid_threshold", "100", "-cluster_fraction", "0.2", "-cluster_iters", "1",
            os.path.join(G, "hemopexin", "hemopexin"), out)
    assert open(out, "rb").read() == open(os.path.join(G, "hemopexin", "medoid-slink-params.dnd"), "rb").read()


def test_cli_medoidtree_duplicates(tmp_path):
    out = str(tmp_path / "t.dnd")
    f = os.path.join(G, "hemopexin_duplicates", "hemopexin_duplicates")
    run_cli("-medoidtree", "-gt", "sl", "-gt_export", f, out)
    assert open(out, "rb").read() == open(os.path.join(G, "hemopexin_duplicates", "medoid-sl.dnd"), "rb").read()
    run_cli("-keep-duplicates", "-medoidtree", "-gt", "sl", "-gt_export", f, out)
    assert open(out, "rb").read() == open(os.path.join(G, "hemopexin_duplicates", "medoid-sl-dups.dnd"), "rb").read()


@pytest.mark.parametrize("batch", ["32", "8", "0"])
@pytest.mark.parametrize("gt", ["upgma", "upgma_modified", "nj"])
def test_device_upgma_on_tie_heavy_inputs(host, tmp_path, monkeypatch, gt, batch):
    """Device UPGMA / NJ vs the host restatement (itself pinned against the reference on tie-heavy random inputs):
    small alphabets give many equal float distances, so every '<' / first-minimum rule is exercised.  UPGMA in batches
    of up to 32 / 8 merges per launch pair (upgma_batch_kernels.hip; with this many equal keys the validity check cuts
    batches short all the time) and with one launch per merge (0)."""
    import numpy as np
    import oracle_bind as ob
    if gt == "nj" and batch != "32":
        pytest.skip("the batch size is a UPGMA parameter")
    monkeypatch.setenv("LCSGPU_UPGMA_BATCH", batch)
    oracle = ob.Oracle()
    for seed, (n, max_len, alpha) in enumerate([(2, 5, "AC"), (3, 8, "AC"), (40, 12, "AC"), (257, 20, "ACD"),
                                                (700, 60, "ACDE"), (1500, 150, "ARNDCQEGHILKMFPSTWYV")]):
        rng = np.random.Generator(np.random.PCG64(500 + seed))
        seqs = ["".join(alpha[i] for i in rng.integers(0, len(alpha), size=int(rng.integers(1, max_len + 1)))) + "A"
                for _ in range(n)]
        fasta = str(tmp_path / f"in{seed}.fasta")
        with open(fasta, "w") as f:
            for i, s in enumerate(seqs):
                f.write(f">r{i}\n{s}\n")
        enc = [oracle.encode(s) for s in seqs]
        codes, offsets = seqio.pack(enc)
        sq = oracle.rect(codes, offsets, np.arange(n), np.arange(n))
        for dist in ("indel075_div_lcs", "indel_div_lcs"):
            want = host.tree_from_matrix(fasta, sq, gt, distance=dist)   # host algorithm over oracle LCS
            got = host.tree_gpu(fasta, gt, distance=dist)                # LCS + UPGMA on the device
            assert got == want, (seed, dist)


@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "nj"])
def test_trees_of_a_set_with_a_giant_sequence(host, oracle, tmp_path, gt):
    """One member beyond 65535 residues: the device reducers run on 32-bit LCS values.  Expected tree =
    the same host layer fed with the oracle's matrix (the CPU suite pins that path to the reference)."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(37))
    lens = [int(x) for x in rng.integers(40, 300, size=60)] + [66500]
    seqs = [rng.integers(0, 20, size=l).astype(np.uint8) for l in lens]
    for i in range(0, 60, 7):
        seqs[-1][i * 300: i * 300 + lens[i]] = seqs[i]  # the giant shares pieces with some members
    codes, offsets = seqio.pack(seqs)
    fasta = str(tmp_path / "giant.fasta")
    seqio.to_fasta(codes, offsets, fasta)
    ids = np.arange(len(seqs), dtype=np.int32)
    square = oracle.rect(codes, offsets, ids, ids)
    assert host.tree_gpu(fasta, gt) == host.tree_from_matrix(fasta, square, gt)


@pytest.mark.parametrize("layout", ["square", "square+b16+spare64", "square+b8+spare32", "square+steps", "triangle"])
@pytest.mark.parametrize("gt", ["upgma", "upgma_modified"])
@pytest.mark.parametrize("shape", ["ties", "family", "hub"])
def test_device_upgma_on_tie_heavy_and_larger_sets(host, oracle, tmp_path, monkeypatch, gt, shape, layout):
    """The device UPGMA against the host restatement (pinned to the reference by the CPU suite) fed with the oracle's
    matrix: thousands of exact distance ties, several workgroups of rows, merges that touch the same workgroup twice
    in a row, a hub every other row is nearest to (one cluster that swallows a row per merge: the chained merges of a
    batch).  Both layouts of the float distances -- the full symmetric matrix (the default while 4 B x n^2 fit; merges
    in batches of 32 / 16 / 8 per launch pair -- also with a spare of 64 / 32 slots only, i.e. a compaction of the live slots
    every other batch --, or one launch per merge) and the packed triangle."""
    import numpy as np
    monkeypatch.setenv("LCSGPU_UPGMA_LAYOUT", layout.split("+")[0])
    if "+spare" in layout:  # a small spare of slots: the live slots are compacted every few batches (upgma_compact_*_kernel)
        monkeypatch.setenv("LCSGPU_TUNE", "upgma_spare=" + layout.split("+spare")[1])
    if "+b" in layout:
        monkeypatch.setenv("LCSGPU_UPGMA_BATCH", layout.split("+b")[1].split("+")[0])
    if layout.endswith("steps"):
        monkeypatch.setenv("LCSGPU_UPGMA_BATCH", "0")
    rng = np.random.Generator(np.random.PCG64(61))
    if shape == "ties":
        seqs = [rng.integers(0, 3, size=int(rng.integers(8, 15))).astype(np.uint8) for _ in range(1300)]
    elif shape == "hub":  # every member is the hub with a few substitutions of its own: the hub is everyone's nearest
        hub = rng.integers(0, 20, size=120, dtype=np.uint8)
        seqs = [hub]
        for _ in range(900):
            s = hub.copy()
            m = rng.random(120) < rng.uniform(0.02, 0.3)
            s[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
            seqs.append(s)
    else:
        anc = rng.integers(0, 20, size=150, dtype=np.uint8)
        seqs = []
        for _ in range(1500):
            s = anc.copy()
            m = rng.random(150) < 0.2
            s[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
            seqs.append(s[: int(rng.integers(90, 151))].copy())
    codes, offsets = seqio.pack(seqs)
    fasta = str(tmp_path / "u.fasta")
    seqio.to_fasta(codes, offsets, fasta)
    ids = np.arange(len(seqs), dtype=np.int32)
    square = oracle.rect(codes, offsets, ids, ids)
    assert host.tree_gpu(fasta, gt, keep_duplicates=True) == host.tree_from_matrix(fasta, square, gt, keep_duplicates=True)


@pytest.mark.parametrize("gt", ["upgma", "upgma_modified"])
def test_device_upgma_forms_agree_where_the_reference_is_undefined(engine, monkeypatch, gt):
    """Two families without a common residue: every distance between them is FLT_MAX (>= UPGMA::BIG_DIST), the reference's
    algorithm is undefined once only such pairs are left (it reads out of bounds).  Whatever the device does -- an error
    that says so, or a tree -- its batched form, its one-launch-per-merge form and a batched form that compacts all the
    time must do the same."""
    import numpy as np
    import famsa_amd
    rng = np.random.Generator(np.random.PCG64(5))
    a = [rng.integers(0, 2, size=int(rng.integers(20, 40))).astype(np.uint8) for _ in range(150)]       # A, R only
    b = [(rng.integers(0, 2, size=int(rng.integers(20, 40))) + 2).astype(np.uint8) for _ in range(120)]  # N, D only
    engine.upload_seqs(a + b)
    outcomes = []
    for env in ({"LCSGPU_UPGMA_BATCH": "32"}, {"LCSGPU_UPGMA_BATCH": "0"}, {"LCSGPU_UPGMA_BATCH": "16", "LCSGPU_TUNE": "upgma_spare=32"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        try:
            left, right = engine.upgma(1, gt == "upgma_modified")
            outcomes.append(("tree", left.tolist(), right.tolist()))
        except famsa_amd.LcsGpuError as e:
            outcomes.append(("error", "no finite nearest neighbour" in str(e)))
        for k in env:
            monkeypatch.delenv(k)
    assert outcomes[0] == outcomes[1] == outcomes[2], [o[0] for o in outcomes]
    # and a set where the rows run out one family at a time but every pair has a finite distance: a tree in all three forms
    engine.upload_seqs([np.concatenate([s, [4]]).astype(np.uint8) for s in a + b])
    trees = []
    for env in ({"LCSGPU_UPGMA_BATCH": "32"}, {"LCSGPU_UPGMA_BATCH": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        left, right = engine.upgma(1, gt == "upgma_modified")
        trees.append((left.tolist(), right.tolist()))
        for k in env:
            monkeypatch.delenv(k)
    assert trees[0] == trees[1]


@pytest.mark.skipif(not oracle_bind.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("heur_id,heur", [(2, "medoidtree"), (1, "parttree")])
def test_deep_recursion_against_the_reference_library(host, tmp_path, monkeypatch, heur_id, heur):
    """A 6000-sequence family with small MedoidTree / PartTree parameters: three levels of recursion with device
    CLARANS searches, seed assignments and batched leaf matrices at every split, eight host threads -- the
    Newick must be the reference library's (its own generators, same parameters)."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(71))
    anc = rng.integers(0, 20, size=120, dtype=np.uint8)
    seqs = []
    for _ in range(6000):
        s = anc.copy()
        m = rng.random(120) < 0.3
        s[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
        seqs.append(s[: int(rng.integers(70, 121))].copy())
    codes, offsets = seqio.pack(seqs)
    fasta = str(tmp_path / "deep.fasta")
    seqio.to_fasta(codes, offsets, fasta)
    monkeypatch.setenv("FAMSA_HOST_TEST", "threads=8")
    ref = oracle_bind.Ref()
    h = ref.open_fasta(fasta)
    try:
        for gt in ("upgma", "sl"):
            want = ref.tree(h, gt, heuristic=heur_id, threads=8, subtree=12, sample=150, threshold=120,
                            cluster_fraction=0.2, cluster_iters=2)
            got = host.tree_gpu(fasta, gt, heuristic=heur, subtree_size=12, sample_size=150, threshold=120,
                                cluster_fraction=0.2, cluster_iters=2)
            assert got == want, (gt, heur)
    finally:
        ref.close(h)


@pytest.mark.parametrize("case,gold", [("adeno_fiber/adeno_fiber", "adeno_fiber/{gt}.dnd"),
                                       ("adversarial_tree.fasta", "adversarial_tree_{gt}.dnd"),
                                       ("hemopexin/hemopexin", "hemopexin/{gt}.dnd")])
@pytest.mark.parametrize("gt", ["sl", "slink"])
def test_single_linkage_without_the_resident_triangle(tmp_path, case, gold, gt):
    """What a set too large for the HBM gets (lcsgpu_mst_prim answers LCSGPU_E_NOMEM): -gt sl by Prim steps that ask
    the engine for one row of the unprocessed vertices each (O(n) memory, MSTPrim::run_view's own structure,
    reference tree/MSTPrim.cpp:356-533), -gt slink by the row-blocked SLINK loop.  Forced here on small inputs."""
    out = str(tmp_path / "t.dnd")
    env = dict(os.environ, FAMSA_HOST_TEST="no_device_mst,prim_streaming")
    p = subprocess.run([host_bind.CLI, "-gt", gt, "-gt_export", os.path.join(G, case), out], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env)
    assert p.returncode == 0, p.stderr
    assert open(out, "rb").read() == open(os.path.join(G, gold.format(gt=gt)), "rb").read()
