"""A REAL, ragged, tie-heavy family beyond 4188 sequences, at size: the upstream real sets held under tests/golden
(adeno_fiber + hemopexin + both duplicates sets + adeno_fiber_extra with its non-standard symbols J / U) as ONE input
of 13 774 records, 21-210 residues, exact duplicates (x2, x3) and thousands of equal distances -- the regime the
integer pre-filter of the MST passes, the tile padding and the half-word-class buckets (4 classes in one launch plan)
are not pinned on by any synthetic set.  Every expectation is the sha256 of the REFERENCE's own output for the same
FASTA (oracle/make_golden_large.py realmix -> tests/golden/meta_large.json; the FASTA is rebuilt here from the
fixtures by the same function and its hash checked first).
Model: the reference's at-size regression on a real family, .github/workflows/self-hosted.yml:395-398, 424-461."""
import hashlib
import json
import os
import subprocess

import pytest

from famsa_amd import seqio
from famsa_amd.hostlib import CLI

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
REC = json.load(open(os.path.join(G, "meta_large.json")))["realmix"]


def file_sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def cli(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([CLI, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return p


@pytest.fixture(scope="module")
def fasta(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("realmix") / "realmix.fasta")
    assert seqio.realmix_fasta(G, path) == REC["n"]
    assert file_sha(path) == REC["fasta_sha256"]
    return path


@pytest.mark.parametrize("keep", [False, True], ids=["unique", "keep-duplicates"])
@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "upgma_modified", "nj"])
def test_trees(fasta, tmp_path, gt, keep):
    if f"{gt}{'_keepdups' if keep else ''}_newick_sha256" not in REC:
        pytest.skip("the reference run is not committed yet (nj -keep-duplicates: O(n^3) on one thread at 13 774 sequences, hours)")
    out = str(tmp_path / "t.dnd")
    cli(*(["-keep-duplicates"] if keep else []), "-gt", gt, "-gt_export", fasta, out)
    assert file_sha(out) == REC[f"{gt}{'_keepdups' if keep else ''}_newick_sha256"]


@pytest.mark.parametrize("gt", ["sl", "upgma"])
def test_trees_with_the_other_distance(fasta, tmp_path, gt):
    """-dist indel_div_lcs on ragged lengths (21-210 aa): the pins differ from the default distance's."""
    if f"{gt}_indel_newick_sha256" not in REC:
        pytest.skip("the reference run is not committed yet (oracle/make_golden_large.py realmixindel)")
    assert REC[f"{gt}_indel_newick_sha256"] != REC[f"{gt}_newick_sha256"]
    out = str(tmp_path / "t.dnd")
    cli("-gt", gt, "-dist", "indel_div_lcs", "-gt_export", fasta, out)
    assert file_sha(out) == REC[f"{gt}_indel_newick_sha256"]


@pytest.mark.parametrize("keep", [False, True], ids=["unique", "keep-duplicates"])
@pytest.mark.parametrize("gt", ["sl", "upgma"])
def test_medoid_trees(fasta, tmp_path, gt, keep):
    out = str(tmp_path / "m.dnd")
    cli(*(["-keep-duplicates"] if keep else []), "-medoidtree", "-gt", gt, "-gt_export", fasta, out)
    assert file_sha(out) == REC[f"medoid_{gt}{'_keepdups' if keep else ''}_newick_sha256"]


def test_dist_export(fasta, tmp_path):
    """854 MB of CSV in the INPUT order (no sort, no duplicate removal: msa.cpp:518-526), byte for byte."""
    out = str(tmp_path / "d.csv")
    cli("-dist_export", fasta, out)
    assert os.path.getsize(out) == REC["dist_export_bytes"]
    assert file_sha(out) == REC["dist_export_sha256"]
    os.unlink(out)


@pytest.mark.parametrize("mode", ["passes", "fused", "recompute"])
def test_single_linkage_in_every_mst_mode(fasta, tmp_path, mode):
    """The three ways the Boruvka rounds get their LCS values (resident triangle + passes, round 0 fused into the LCS
    launch, no triangle at all), with duplicates kept: 9 000 exact-duplicate pairs at distance 0 and their ties."""
    out = str(tmp_path / "s.dnd")
    cli("-keep-duplicates", "-gt", "sl", "-gt_export", fasta, out, env={"LCSGPU_MST_MODE": mode})
    assert file_sha(out) == REC["sl_keepdups_newick_sha256"]


@pytest.mark.parametrize("gt", ["sl", "upgma", "nj"])
def test_two_contexts(fasta, tmp_path, gt):
    """Row blocks on two contexts (`-gpu 0,0`): the sharded Boruvka rounds / the gathered triangle of the matrix consumers."""
    out = str(tmp_path / "g.dnd")
    cli("-gpu", "0,0", "-gt", gt, "-gt_export", fasta, out)
    assert file_sha(out) == REC[f"{gt}_newick_sha256"]
    if gt == "sl":
        cli("-gpu", "0,0", "-keep-duplicates", "-gt", gt, "-gt_export", fasta, out)
        assert file_sha(out) == REC["sl_keepdups_newick_sha256"]


@pytest.mark.parametrize("batch", ["32", "8", "0"])
def test_upgma_batch_sizes(fasta, tmp_path, batch):
    """UPGMA with duplicates kept -- thousands of rows whose minimum is 0 and equal: the validity check of the batched
    merges (upgma_batch_kernels.hip) cuts batches short here -- in batches of 32 / 8 and with one launch per merge."""
    out = str(tmp_path / "u.dnd")
    cli("-keep-duplicates", "-gt", "upgma", "-gt_export", fasta, out, env={"LCSGPU_UPGMA_BATCH": batch})
    assert file_sha(out) == REC["upgma_keepdups_newick_sha256"]
