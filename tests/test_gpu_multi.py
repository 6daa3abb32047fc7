"""Several GPUs on one problem, emulated on the 1-GPU box by several contexts on cuda:0 (`-gpu 0,0`): the
lcsgpu_multi_* calls of include/lcsgpu.h (row blocks of equal pair counts per context; device-to-device
gather of the blocks for UPGMA / NJ; Boruvka with the keys exchanged through host memory for single linkage)
and the famsa-gpu host layer on top (leaf batches split by pair count, seed assignment split by columns,
small requests round robin).  Everything must be byte-identical to the one-context results / the goldens."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import famsa_amd
from famsa_amd import seqio
from famsa_amd.hostlib import CLI
from famsa_amd.lcsgpu import MST_TRIANGLE_ORIENTATION

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(G, "meta.json")))
META_LARGE = json.load(open(os.path.join(G, "meta_large.json")))


def cli(*args):
    p = subprocess.run([CLI, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return p


def file_sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def _sets():
    rng = np.random.Generator(np.random.PCG64(91))
    ties = [rng.integers(0, 3, size=int(rng.integers(4, 14))).astype(np.uint8) for _ in range(1800)]
    fam = seqio.synth_family(2500, 150, seed=3)
    return {"ties": ties, "family": fam}


@pytest.mark.parametrize("name", ["ties", "family"])
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_group_calls_equal_single_context(engine, name, devices):
    seqs = _sets()[name]
    engine.upload_seqs(seqs)
    grp = famsa_amd.LcsGpuGroup(devices)
    try:
        grp.upload_seqs(seqs)
        n = len(seqs)
        assert (grp.lcs_triangle() == engine.lcs_triangle()).all()
        r0, r1 = n // 3, n - 5
        assert (grp.lcs_triangle(r0, r1) == engine.lcs_triangle(r0, r1)).all()
        for kind in (1, 0, 1 | MST_TRIANGLE_ORIENTATION):
            a, b = grp.mst_prim(kind), engine.mst_prim(kind)
            assert (a["from"] == b["from"]).all() and (a["to"] == b["to"]).all()
            assert (a["dist"].view(np.uint64) == b["dist"].view(np.uint64)).all()
        if name == "family":  # (tie-quantised toy distances make the reference's UPGMA/NJ input degenerate)
            for modified in (False, True):
                gl, gr = grp.upgma(1, modified)
                sl, sr = engine.upgma(1, modified)
                assert (gl == sl).all() and (gr == sr).all()
            gl, gr = grp.nj(1)
            sl, sr = engine.nj(1)
            assert (gl == sl).all() and (gr == sr).all()
    finally:
        grp.close()


@pytest.mark.parametrize("switch", ["peer", "host"])
def test_group_calls_over_the_other_transports(engine, monkeypatch, switch):
    """Contexts on ONE device copy with hipMemcpyAsync; these switches make the same calls take the branches two
    devices take: hipMemcpyPeerAsync (with peer access: xGMI) resp. the pinned-host staging used when two devices
    cannot address each other -- the row-block gather of lcsgpu_multi_upgma and the per-round key pushes of
    lcsgpu_multi_mst_prim."""
    monkeypatch.setenv("LCSGPU_TRANSPORT", switch)
    seqs = _sets()["family"]
    engine.upload_seqs(seqs)
    grp = famsa_amd.LcsGpuGroup([0, 0, 0])
    try:
        grp.upload_seqs(seqs)
        a, b = grp.mst_prim(1), engine.mst_prim(1)
        assert (a["from"] == b["from"]).all() and (a["to"] == b["to"]).all()
        assert (a["dist"].view(np.uint64) == b["dist"].view(np.uint64)).all()
        gl, gr = grp.upgma(1, False)
        sl, sr = engine.upgma(1, False)
        assert (gl == sl).all() and (gr == sr).all()
    finally:
        grp.close()


def test_key_exchange_as_an_rccl_all_gather(engine, monkeypatch):
    """LCSGPU_EXCHANGE=rccl: the per-round key exchange of lcsgpu_multi_mst_prim is one grouped ncclAllGather on the
    contexts' streams (librccl loaded on demand, ncclCommInitAll over the contexts' devices).  RCCL wants one device
    per rank, so what a 1-GPU box can run is the group of ONE context -- communicator, in-place all-gather and
    stream ordering all execute (with this switch the call does not shortcut to lcsgpu_mst_prim) -- and the refusal
    of contexts that share a device."""
    monkeypatch.setenv("LCSGPU_EXCHANGE", "rccl")
    seqs = _sets()["family"]
    engine.upload_seqs(seqs)
    one = famsa_amd.LcsGpuGroup([0])
    two = famsa_amd.LcsGpuGroup([0, 0])
    try:
        one.upload_seqs(seqs)
        for kind in (1, 0):
            a = one.mst_prim(kind)
            monkeypatch.delenv("LCSGPU_EXCHANGE")
            b = engine.mst_prim(kind)
            monkeypatch.setenv("LCSGPU_EXCHANGE", "rccl")
            assert (a["from"] == b["from"]).all() and (a["to"] == b["to"]).all()
            assert (a["dist"].view(np.uint64) == b["dist"].view(np.uint64)).all()
        two.upload_seqs(seqs)
        with pytest.raises(famsa_amd.LcsGpuError, match="share device"):
            two.mst_prim(1)
    finally:
        one.close()
        two.close()


def test_exchange_choice_cross_check_and_transport_report(engine, monkeypatch, capfd):
    """Which exchange lcsgpu_multi_mst_prim picks and what lcsgpu_multi_transport says about it.  On one device the
    automatic choice must be the peer copies (RCCL wants one device per rank) and say so; LCSGPU_EXCHANGE=rccl on one
    context runs the all-gather WITH the first-use cross-check against the peer-copy form (LCSGPU_EXCHANGE_CHECK=always
    repeats it); the per-context kernel times of the call are all readable afterwards."""
    seqs = _sets()["family"]
    engine.upload_seqs(seqs)
    want = engine.mst_prim(1)
    grp = famsa_amd.LcsGpuGroup([0, 0, 0])
    one = famsa_amd.LcsGpuGroup([0])
    try:
        grp.upload_seqs(seqs)
        got = grp.mst_prim(1)
        assert (got["from"] == want["from"]).all() and (got["to"] == want["to"]).all()
        text = grp.transport()
        assert "contexts: 3 on devices 0 0 0" in text and "same-device x6" in text, text
        assert "peer copies (automatic: contexts share a device" in text, text
        ms = grp.last_kernel_ms()
        assert len(ms) == 3 and all(m > 0 for m in ms), ms
        monkeypatch.setenv("LCSGPU_TRANSPORT", "host")
        assert "host-staging(forced) x6" in grp.transport() and "not by peer copy: 0->1" in grp.transport()
        monkeypatch.delenv("LCSGPU_TRANSPORT")
        monkeypatch.setenv("LCSGPU_EXCHANGE", "peer")
        grp.mst_prim(1)
        assert "peer copies (LCSGPU_EXCHANGE=peer)" in grp.transport()
        # the RCCL form with its cross-check, on the one shape a 1-GPU box can give RCCL
        one.upload_seqs(seqs)
        monkeypatch.setenv("LCSGPU_EXCHANGE", "rccl")
        monkeypatch.setenv("LCSGPU_EXCHANGE_CHECK", "always")
        monkeypatch.setenv("LCSGPU_PROFILE", "1")
        got = one.mst_prim(1)
        err = capfd.readouterr().err
        assert "first round cross-checked against peer copies" in err, err
        assert (got["from"] == want["from"]).all() and (got["dist"].view(np.uint64) == want["dist"].view(np.uint64)).all()
        assert "rccl: one grouped ncclAllGather per round" in one.transport()
    finally:
        grp.close()
        one.close()


def test_row_blocks_come_back_over_all_links_at_once(engine, monkeypatch):
    """lcsgpu_multi_lcs_triangle drains every context's block on its own host thread (one PCIe link per GPU, all at the
    same time): the same bytes as one context's triangle."""
    seqs = _sets()["family"]
    engine.upload_seqs(seqs)
    want = engine.lcs_triangle()
    grp = famsa_amd.LcsGpuGroup([0, 0, 0, 0])
    try:
        grp.upload_seqs(seqs)
        assert (grp.lcs_triangle() == want).all()
    finally:
        grp.close()


def test_group_refuses_what_it_cannot_do(engine):
    seqs = [np.zeros(192, np.uint8)] + _sets()["family"][:300]  # a carry-quirk (orientation-sensitive) sequence
    grp = famsa_amd.LcsGpuGroup([0, 0])
    other = famsa_amd.LcsGpu(0)
    try:
        grp.upload_seqs(seqs)
        with pytest.raises(famsa_amd.LcsGpuError, match="orientation sensitive"):
            grp.mst_prim(1)
        engine.upload_seqs(seqs)
        a, b = grp.mst_prim(1 | MST_TRIANGLE_ORIENTATION), engine.mst_prim(1 | MST_TRIANGLE_ORIENTATION)
        assert (a["from"] == b["from"]).all() and (a["to"] == b["to"]).all()
        grp.engs[1].upload_seqs(seqs[:100])  # the contexts no longer hold the same set
        with pytest.raises(famsa_amd.LcsGpuError, match="different sequence set"):
            grp.lcs_triangle()
    finally:
        grp.close()
        other.close()


@pytest.mark.parametrize("gt", ["sl", "slink", "upgma", "nj"])
@pytest.mark.parametrize("case", ["adeno_fiber/adeno_fiber", "adversarial_tree.fasta", "hemopexin/hemopexin"])
def test_cli_two_contexts_trees(tmp_path, case, gt):
    """adversarial_tree holds orientation-sensitive sequences: -gt sl falls back to one context there."""
    gold = {"adeno_fiber/adeno_fiber": f"adeno_fiber/{gt}.dnd", "adversarial_tree.fasta": f"adversarial_tree_{gt}.dnd",
            "hemopexin/hemopexin": f"hemopexin/{gt}.dnd"}[case]
    out = str(tmp_path / "t.dnd")
    cli("-gpu", "0,0", "-gt", gt, "-gt_export", os.path.join(G, case), out)
    assert open(out, "rb").read() == open(os.path.join(G, gold), "rb").read()


def test_cli_verbose_reports_the_transport(tmp_path):
    """`famsa-gpu -v -gpu a,b`: the `gpu.transport=` lines of lcsgpu_multi_transport -- how the contexts reach each other and
    which key exchange the single-linkage call used (on one device: peer copies, and why)."""
    out = str(tmp_path / "t.dnd")
    p = cli("-v", "-gpu", "0,0", "-gt", "sl", "-gt_export", os.path.join(G, "hemopexin", "hemopexin"), out)
    lines = [l for l in p.stderr.splitlines() if l.startswith("gpu.transport=")]
    assert any("contexts: 2 on devices 0 0" in l for l in lines), p.stderr[-1500:]
    assert any("same-device x2" in l for l in lines)
    assert any("key exchange of the last single-linkage call: peer copies (automatic" in l for l in lines)
    assert open(out, "rb").read() == open(os.path.join(G, "hemopexin", "sl.dnd"), "rb").read()


def test_cli_three_contexts_dist_export(tmp_path):
    out = str(tmp_path / "d.csv")
    cli("-gpu", "0,0,0", "-dist_export", os.path.join(G, "hemopexin", "hemopexin"), out)
    assert os.path.getsize(out) == META["hemopexin"]["dist_csv_bytes"]
    assert file_sha(out) == META["hemopexin"]["dist_csv_sha256"]
    out2 = str(tmp_path / "sq.csv")
    cli("-gpu", "0,0", "-dist_export", "-square_matrix", os.path.join(G, "adeno_fiber", "adeno_fiber"), out2)
    assert open(out2, "rb").read() == open(os.path.join(G, "adeno_fiber", "dist_sq.csv"), "rb").read()


def test_cli_medoid_trees_over_two_contexts(tmp_path):
    for gt in ("sl", "upgma"):
        out = str(tmp_path / f"m_{gt}.dnd")
        cli("-gpu", "0,0", "-medoidtree", "-gt", gt, "-gt_export", os.path.join(G, "hemopexin", "hemopexin"), out)
        assert open(out, "rb").read() == open(os.path.join(G, "hemopexin", f"medoid-{gt}.dnd"), "rb").read()
    rec = META_LARGE["family200000"]
    path = str(tmp_path / "family.fasta")
    seqio.family_fasta(200000, rec["len"], path)
    out = str(tmp_path / "medoid.dnd")
    cli("-gpu", "0,0,0", "-medoidtree", "-gt", "upgma", "-gt_export", path, out)
    assert file_sha(out) == rec["medoid_upgma_newick_sha256"]


def test_cli_c4_over_three_contexts(tmp_path):
    codes, offsets = seqio.synth_uniform(100000, 400)
    path = str(tmp_path / "synth100k.fasta")
    seqio.to_fasta(codes, offsets, path)
    out = str(tmp_path / "sl.dnd")
    p = cli("-v", "-gpu", "0,0,0", "-gt", "sl", "-gt_export", path, out)
    assert file_sha(out) == META_LARGE["synth100k"]["sl_newick_sha256"], p.stderr
