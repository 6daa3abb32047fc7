"""BASELINE configs at their size on the GPU, against sha256 values produced once by the REFERENCE itself
(oracle/make_golden_large.py -> tests/golden/meta_large.json; the inputs are regenerated here by the same
deterministic generators).  Model: the reference's own at-size regression, .github/workflows/self-hosted.yml:424-461.

  C3  synthetic 10 000 x 400 aa: the u16 LCS triangle and the sl / slink / upgma / upgma_modified / nj Newick;
      -dist indel_div_lcs: sl / upgma (round 5)
  C4  synthetic 100 000 x 400 aa: -gt sl Newick (one GPU; the 2-context row-block form too), sampled oracle check;
      -gt upgma / upgma_modified Newick; round 5: -gt slink, and -dist indel_div_lcs with sl / upgma
  C5  'family' sets of 200 000 and 1 000 000 sequences: -medoidtree -gt upgma Newick
      and of 3 000 000 sequences (the C5 shape at its size); FAMSA_TEST_HUGE=1 adds the host-CLARANS / 1-thread run
"""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import famsa_amd
from famsa_amd import seqio
from famsa_amd.hostlib import CLI

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
META = json.load(open(os.path.join(ROOT, "tests", "golden", "meta_large.json")))


def sha(b):
    return hashlib.sha256(b).hexdigest()


def file_sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def cli(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([CLI, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    return p


@pytest.fixture(scope="module")
def synth10k(tmp_path_factory):
    codes, offsets = seqio.synth_uniform(10000, 400)
    assert sha(codes.tobytes()) == META["synth10k"]["codes_sha256"]
    path = str(tmp_path_factory.mktemp("c3") / "synth10k.fasta")
    seqio.to_fasta(codes, offsets, path)
    return codes, offsets, path


@pytest.fixture(scope="module")
def synth100k(tmp_path_factory):
    codes, offsets = seqio.synth_uniform(100000, 400)
    assert sha(codes.tobytes()) == META["synth100k"]["codes_sha256"]
    path = str(tmp_path_factory.mktemp("c4") / "synth100k.fasta")
    seqio.to_fasta(codes, offsets, path)
    return codes, offsets, path


def test_c3_triangle(engine, synth10k):
    codes, offsets, _ = synth10k
    engine.upload(codes, offsets)  # fixed length, generator order: the reference's rows are the same sequences
    assert sha(engine.lcs_triangle().tobytes()) == META["synth10k"]["triangle_u16_sha256"]


@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "upgma_modified", "nj"])
def test_c3_trees(synth10k, gt):
    """(upgma_modified and nj were added in round 3 together with -ffp-contract=off: both reducers have a
    multiply-add the compiler used to fuse.)"""
    got = famsa_amd.guide_tree(synth10k[2], gt)
    assert sha(got) == META["synth10k"][f"{gt}_newick_sha256"]


def pinned(section, key):
    """The reference's sha256, or a skip that says which generator run is missing (oracle/make_golden_large.py)."""
    if key not in META.get(section, {}):
        pytest.skip(f"tests/golden/meta_large.json has no {section}.{key}: the reference run (oracle/make_golden_large.py) is not committed yet")
    return META[section][key]


@pytest.mark.parametrize("gt", ["sl", "upgma"])
def test_c3_trees_with_the_other_distance(synth10k, gt):
    """-dist indel_div_lcs (Transform<.., indel_div_lcs>, reference tree/AbstractTreeGenerator.hpp:65-75) at C3's size."""
    got = famsa_amd.guide_tree(synth10k[2], gt, distance="indel_div_lcs")
    assert sha(got) == pinned("synth10k", f"{gt}_indel_newick_sha256")


@pytest.mark.parametrize("gt,dist", [("slink", "indel075_div_lcs"), ("sl", "indel_div_lcs"), ("upgma", "indel_div_lcs")])
def test_c4_trees_pinned_in_round_5(synth100k, tmp_path, gt, dist):
    """-gt slink at 100 000 sequences (the device MST over SLINK's orientation + MST -> pointer representation, against the
    reference's sequential SLINK: tree/SingleLinkage.cpp:31-189) and the other distance measure with sl / upgma."""
    want = pinned("synth100k", f"{gt}{'_indel' if dist == 'indel_div_lcs' else ''}_newick_sha256")
    out = str(tmp_path / f"{gt}.dnd")
    cli("-gt", gt, "-dist", dist, "-gt_export", synth100k[2], out)
    assert file_sha(out) == want


def test_c4_single_linkage_tree(synth100k, tmp_path):
    out = str(tmp_path / "sl.dnd")
    cli("-gt", "sl", "-gt_export", synth100k[2], out)
    assert file_sha(out) == META["synth100k"]["sl_newick_sha256"]


@pytest.mark.parametrize("gt,layout", [("upgma", "square+steps"), ("upgma", "triangle"), ("upgma_modified", "square")])
def test_c4_upgma_trees(synth100k, tmp_path, gt, layout):
    """100 000 merges on the device (one launch each) over the float distances -- the 40 GB symmetric matrix (default)
    or the 20 GB packed triangle: the per-workgroup minima are two per thread at this size (391 workgroups), which no
    smaller case reaches.  Against the sha256 of the REFERENCE's own runs (oracle/make_golden_large.py c4upgma)."""
    out = str(tmp_path / f"{gt}.dnd")
    env = {"LCSGPU_UPGMA_LAYOUT": layout.split("+")[0]}
    if layout.endswith("steps"):  # one launch per merge (the default until round 4: batches of 32 merges per launch pair)
        env["LCSGPU_UPGMA_BATCH"] = "0"
    cli("-gt", gt, "-gt_export", synth100k[2], out, env=env)
    assert file_sha(out) == META["synth100k"][f"{gt}_newick_sha256"]


def test_c4_row_block_contexts_tree_and_sampled_triangle(oracle, synth100k):
    """C4's shape on one GPU: two contexts holding the two row blocks of the 100 000-sequence triangle (5 GB
    each), Boruvka rounds with the key exchange in device memory; 6000 sampled pairs against the oracle; the
    edges must be the single-context MST's (whose Newick the test above pins to the reference)."""
    import torch
    from famsa_amd.rowblock import row_cuts, pairs_in_rows, edge_list_sha256
    codes, offsets, _ = synth100k
    n = 100000
    cuts = row_cuts(n, 2)
    engs, tris = [], []
    for p in range(2):
        e = famsa_amd.LcsGpu(0)
        e.upload(codes, offsets)
        t = torch.empty(pairs_in_rows(cuts[p], cuts[p + 1]), dtype=torch.int16, device="cuda:0")
        e.lcs_triangle_dev(cuts[p], cuts[p + 1], t.data_ptr(), 2, sync=True)
        engs.append(e)
        tris.append(t)
    try:
        rng = np.random.Generator(np.random.PCG64(4))
        o = offsets.astype(np.int64)
        for p in range(2):
            r0, r1 = cuts[p], cuts[p + 1]
            rows = rng.integers(max(r0, 1), r1, size=3000)
            cols = (rng.random(3000) * rows).astype(np.int64)
            idx = rows * (rows - 1) // 2 + cols - r0 * (r0 - 1) // 2
            got = tris[p][torch.from_numpy(idx).cuda()].cpu().numpy().astype(np.int64) & 0xFFFF
            for k in range(3000):
                i, j = int(rows[k]), int(cols[k])
                assert got[k] == oracle.lcs(codes[o[i]:o[i + 1]], codes[o[j]:o[j + 1]]), (i, j)
        for p, e in enumerate(engs):
            e.mst_shard_begin(tris[p].data_ptr(), 2, cuts[p], cuts[p + 1], 1)
        keys = [torch.zeros(2 * n, dtype=torch.int64, device="cuda:0") for _ in engs]
        torch.cuda.synchronize()  # torch fills them on ITS stream; the engines write them on theirs
        found = 0
        while found < n - 1:
            for e, k in zip(engs, keys):
                e.mst_shard_best(k.data_ptr())
            for e in engs:
                e.sync()
            gathered = torch.cat(keys)
            torch.cuda.synchronize()
            found = [e.mst_shard_merge(gathered.data_ptr(), 2) for e in engs][0]
        sharded = [e.mst_shard_finish() for e in engs]
        del tris, keys, gathered
        torch.cuda.empty_cache()
        single = engs[0].mst_prim(1)
        assert edge_list_sha256(sharded[0]) == edge_list_sha256(sharded[1]) == edge_list_sha256(single)
    finally:
        for e in engs:
            e.close()


@pytest.mark.parametrize("n,env", [(200000, {}), (1000000, {}), (200000, {"LCSGPU_TUNE": "clarans_slice_us=150,clarans_draws=500,assign_batch_kb=4096,narrow_lists=0"}),
                                   (200000, {"FAMSA_HOST_TEST": "release_early,no_spare_tree,no_level_scratch"})],
                         ids=["200000", "1000000", "200000-short-slices", "200000-host-memory-as-it-was"])
def test_c5_medoid_tree(tmp_path, n, env):
    """(the last case: the residues released beside the levels, no tree set up during the upload, the levels' arrays allocated
    anew -- how the host side held its memory before the second session of round 6; where memory comes from changes no tree)"""
    rec = META[f"family{n}"]
    path = str(tmp_path / f"family_{n}.fasta")
    seqio.family_fasta(n, rec["len"], path)
    assert file_sha(path) == rec["fasta_sha256"]
    out = str(tmp_path / "medoid.dnd")
    cli("-medoidtree", "-gt", "upgma", "-gt_export", path, out, env=env)
    assert os.path.getsize(out) == rec["newick_bytes"]
    assert file_sha(out) == rec["medoid_upgma_newick_sha256"]


@pytest.mark.parametrize("gt,dist", [("sl", "indel_div_lcs"), ("upgma", "indel_div_lcs"), ("sl", "indel075_div_lcs")])
def test_ragged_100000_with_both_distances(tmp_path, gt, dist):
    """-dist indel_div_lcs where it DISCRIMINATES: at one fixed length (synth10k / synth100k) both Transforms are monotone
    in the LCS length, so single linkage gives the same tree for either and those sl_indel pins equal the sl pins by
    construction; on the ragged family set (lengths 210-300) the two order the pairs differently
    (reference tree/AbstractTreeGenerator.hpp:65-75; the pins: oracle/make_golden_large.py familyindel)."""
    tag = "_indel" if dist == "indel_div_lcs" else ""
    want = pinned("family100000", f"{gt}{tag}_newick_sha256")
    if "sl_indel_newick_sha256" in META.get("family100000", {}) and "sl_newick_sha256" in META["family100000"]:
        assert META["family100000"]["sl_indel_newick_sha256"] != META["family100000"]["sl_newick_sha256"]  # it does discriminate
    path = str(tmp_path / "family_100000.fasta")
    seqio.family_fasta(100000, 300, path)
    out = str(tmp_path / "t.dnd")
    cli("-gt", gt, "-dist", dist, "-gt_export", path, out)
    assert file_sha(out) == want


def test_c5_level_batches_against_split_by_split(tmp_path):
    """The level-by-level walk with the engine's batched calls (lcsgpu_clarans_batch, lcsgpu_assign_seeds_batch) and the same
    walk asking split by split (lcsgpu_clarans, lcsgpu_assign_seeds): one tree, the reference's."""
    rec = META["family200000"]
    path = str(tmp_path / "family_200000.fasta")
    seqio.family_fasta(200000, rec["len"], path)
    out = str(tmp_path / "split_by_split.dnd")
    cli("-medoidtree", "-gt", "upgma", "-gt_export", path, out, env={"FAMSA_HOST_TEST": "no_level_batch"})
    assert file_sha(out) == rec["medoid_upgma_newick_sha256"]


def test_c5_three_million(tmp_path):
    """BASELINE config C5 at its size: -medoidtree -gt upgma over 3 000 000 sequences, against the sha256 of the
    REFERENCE's own run (oracle/make_golden_large.py c5huge: 444 s on the build container's 8 cores).  With
    FAMSA_TEST_HUGE=1 additionally: the plainest path this engine has (host CLARANS, one host thread) must give the
    same file -- the two paths share only the LCS kernels."""
    rec = META["family3000000"]
    path = str(tmp_path / "family_3m.fasta")
    seqio.family_fasta(rec["n"], rec["len"], path)
    assert file_sha(path) == rec["fasta_sha256"]
    a = str(tmp_path / "a.dnd")
    cli("-medoidtree", "-gt", "upgma", "-gt_export", path, a)
    assert os.path.getsize(a) == rec["newick_bytes"]
    assert file_sha(a) == rec["medoid_upgma_newick_sha256"]
    if os.environ.get("FAMSA_TEST_HUGE"):
        b = str(tmp_path / "b.dnd")
        cli("-medoidtree", "-gt", "upgma", "-t", "1", "-gt_export", path, b, env={"FAMSA_HOST_TEST": "clarans_host"})
        assert file_sha(b) == rec["medoid_upgma_newick_sha256"]
