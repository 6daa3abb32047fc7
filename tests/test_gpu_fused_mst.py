"""The local half of a Boruvka round FUSED into the LCS launch (include/lcsgpu.h, LCSGPU_MST_COMPUTE;
lcs_kernels.hip, FuseArgs): the workgroup that holds LCS(row, 256 columns) in registers folds it into every vertex's
best edge before (or instead of) storing it.  Three ways to the same tree, which must agree bit for bit --
  passes     the triangle in HBM, every round by the streaming passes (round 2's form),
  fused      the triangle in HBM, round 0 done by the launch that fills it,
  recompute  NO triangle (O(n) memory, what lifts the n <= ~530 000 limit): every round recomputes the LCS values --
on the shapes that stress the record / filter logic: ties everywhere, variable lengths over several half-word
classes, orientation-sensitive refs (the quirk kernels) in the triangle orientation, refs beyond 2048 residues (the
long kernel), empty sequences, LCS-0 pairs, row-block shards."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import famsa_amd
from famsa_amd import seqio
from famsa_amd.hostlib import CLI
from famsa_amd.lcsgpu import MST_COMPUTE, MST_TRIANGLE_ORIENTATION
from famsa_amd.rowblock import row_cuts, pairs_in_rows

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
META_LARGE = json.load(open(os.path.join(G, "meta_large.json")))


def same(a, b):
    return (a["from"] == b["from"]).all() and (a["to"] == b["to"]).all() and \
        (a["dist"].view(np.uint64) == b["dist"].view(np.uint64)).all()


def _sets():
    rng = np.random.Generator(np.random.PCG64(17))
    out = {}
    out["ties"] = [rng.integers(0, 3, size=int(rng.integers(4, 14))).astype(np.uint8) for _ in range(1500)]
    out["family"] = seqio.synth_family(3000, 150, seed=5)
    # several half-word classes in one set (lengths 20 .. 700), working order (length descending) and shuffled
    mixed = [rng.integers(0, 20, size=int(l)).astype(np.uint8) for l in rng.integers(20, 700, size=1200)]
    out["mixed_sorted"] = [mixed[i] for i in seqio.sort_order(mixed)]
    out["mixed_shuffled"] = mixed
    # orientation-sensitive refs (an aligned 64-residue homopolymer word at word index >= 1), empty and all-X sequences
    quirk = [np.zeros(192, np.uint8), np.full(130, 3, np.uint8), np.zeros(0, np.uint8), np.full(9, 22, np.uint8)]
    quirk += [np.concatenate([rng.integers(0, 20, size=64), np.full(64, 7), rng.integers(0, 20, size=int(k))]).astype(np.uint8)
              for k in rng.integers(0, 90, size=40)]
    out["quirk"] = quirk + seqio.synth_family(500, 120, seed=9)
    # refs beyond 2048 residues next to short ones
    out["long"] = [rng.integers(0, 20, size=int(l)).astype(np.uint8) for l in [2100, 3000, 2500, 2049, 4200] * 4] + \
        seqio.synth_family(300, 200, seed=2)
    return out


@pytest.mark.parametrize("name", ["ties", "family", "mixed_sorted", "mixed_shuffled", "quirk", "long"])
def test_three_ways_to_the_same_tree(engine, monkeypatch, name):
    seqs = _sets()[name]
    engine.upload_seqs(seqs)
    kinds = [1 | MST_TRIANGLE_ORIENTATION, 0 | MST_TRIANGLE_ORIENTATION]
    if name not in ("quirk",):
        kinds.append(1)  # MSTPrim's own orientation (no orientation-sensitive sequence in these sets)
    for kind in kinds:
        monkeypatch.setenv("LCSGPU_MST_MODE", "passes")
        want = engine.mst_prim(kind)
        for mode in ("fused", "recompute"):
            monkeypatch.setenv("LCSGPU_MST_MODE", mode)
            got = engine.mst_prim(kind)
            assert same(got, want), (name, kind, mode)
    monkeypatch.delenv("LCSGPU_MST_MODE")
    assert same(engine.mst_prim(kinds[0]), engine.mst_prim(kinds[0]))


def test_no_room_for_the_triangle_means_recompute(engine, monkeypatch):
    """The automatic choice: when 2 B per pair do not fit the free device memory, lcsgpu_mst_prim keeps no triangle."""
    seqs = _sets()["family"]
    engine.upload_seqs(seqs)
    want = engine.mst_prim(1)
    probe = famsa_amd.LcsGpu(0)  # a fresh context: no result buffer of an earlier call that would already fit
    try:
        probe.upload_seqs(seqs)
        monkeypatch.setenv("LCSGPU_FAKE_HBM_GB", "0.001")  # 1 MB "free": the 9 MB triangle does not fit
        assert same(probe.mst_prim(1), want)
        with pytest.raises(famsa_amd.LcsGpuError, match="device memory"):
            probe.upgma(1)  # the matrix consumers do need their triangle: a clean LCSGPU_E_NOMEM
    finally:
        probe.close()


def test_part_of_the_triangle_resident(engine, monkeypatch, capfd):
    """Between "fits" and "nothing fits": the rows that fit stay in HBM (passes), the rest is recomputed per round
    (fused launch); the two blocks' keys are merged like two ranks' keys."""
    seqs = _sets()["mixed_sorted"] + _sets()["family"]
    seqs = [seqs[i] for i in seqio.sort_order(seqs)]
    engine.upload_seqs(seqs)
    n = len(seqs)
    want = engine.mst_prim(1 | MST_TRIANGLE_ORIENTATION)
    probe = famsa_amd.LcsGpu(0)
    try:
        probe.upload_seqs(seqs)
        monkeypatch.setenv("LCSGPU_PROFILE", "1")
        monkeypatch.setenv("LCSGPU_FAKE_HBM_GB", "%.6f" % (0.6 * n * (n - 1) / 1e9))  # room for ~60 % of the 2 B x pairs
        capfd.readouterr()
        got = probe.mst_prim(1 | MST_TRIANGLE_ORIENTATION)
        err = capfd.readouterr().err
        assert "hybrid" in err, err
        assert same(got, want)
    finally:
        probe.close()


@pytest.mark.parametrize("parts", [2, 3])
@pytest.mark.parametrize("resident", [True, False])
def test_row_block_shards_with_the_fold_in_the_launch(engine, parts, resident):
    """lcsgpu_mst_shard_begin with LCSGPU_MST_COMPUTE: `parts` contexts over disjoint row blocks -- the triangle of a
    block filled by the fused launch (resident) or never stored (recompute) -- keys exchanged in device memory."""
    import torch
    seqs = _sets()["mixed_sorted"]
    engine.upload_seqs(seqs)
    want = engine.mst_prim(1)
    n = len(seqs)
    cuts = row_cuts(n, parts)
    engs, tris = [], []
    try:
        for p in range(parts):
            e = famsa_amd.LcsGpu(0)
            e.upload_seqs(seqs)
            t = torch.empty(max(pairs_in_rows(cuts[p], cuts[p + 1]), 1), dtype=torch.int16, device="cuda:0") if resident else None
            engs.append(e)
            tris.append(t)
        torch.cuda.synchronize()
        for p, e in enumerate(engs):
            e.mst_shard_begin(tris[p].data_ptr() if resident else None, 2, cuts[p], cuts[p + 1], 1 | MST_COMPUTE)
        keys = [torch.zeros(2 * n, dtype=torch.int64, device="cuda:0") for _ in engs]
        torch.cuda.synchronize()
        found, rounds = 0, 0
        while found < n - 1:
            assert rounds < 40
            for e, k in zip(engs, keys):
                e.mst_shard_best(k.data_ptr())
            for e in engs:
                e.sync()
            gathered = torch.cat(keys)
            torch.cuda.synchronize()
            found = [e.mst_shard_merge(gathered.data_ptr(), parts) for e in engs][0]
            rounds += 1
        for e in engs:
            assert same(e.mst_shard_finish(), want)
        if resident:  # the fused launch stored the block's triangle as lcsgpu_lcs_triangle_dev does
            for p in range(parts):
                got = tris[p].cpu().numpy().view(np.uint16)[: pairs_in_rows(cuts[p], cuts[p + 1])]
                assert (got == engine.lcs_triangle(cuts[p], cuts[p + 1])).all()
    finally:
        for e in engs:
            e.close()


@pytest.mark.parametrize("parts", [1, 3])
@pytest.mark.parametrize("resident", [True, False])
def test_host_merge_protocol_with_the_fold_in_the_launch(engine, parts, resident):
    """LCSGPU_MST_COMPUTE with the HOST form of the protocol include/lcsgpu.h documents: shard_best (host keys) ->
    lcsgpu_mst_merge_host -> lcsgpu_mst_shard_set_components.  The round state must advance in set_components too:
    from round 1 on the fused launches fold with the labels (no triangle) resp. the passes replace the spent round-0
    records (resident triangle) -- else the keys name edges inside a component and the merge refuses them."""
    import torch
    from famsa_amd.rowblock import sharded_mst_host
    seqs = _sets()["mixed_sorted"]
    engine.upload_seqs(seqs)
    want = engine.mst_prim(1)
    n = len(seqs)
    cuts = row_cuts(n, parts)
    engs, tris = [], []
    try:
        for p in range(parts):
            e = famsa_amd.LcsGpu(0)
            e.upload_seqs(seqs)
            engs.append(e)
            tris.append(torch.empty(max(pairs_in_rows(cuts[p], cuts[p + 1]), 1), dtype=torch.int16, device="cuda:0") if resident else None)
        torch.cuda.synchronize()
        for p, e in enumerate(engs):
            e.mst_shard_begin(tris[p].data_ptr() if resident else None, 2, cuts[p], cuts[p + 1], 1 | MST_COMPUTE)

        def set_components(comp):
            for e in engs:
                e.mst_shard_set_components(comp)

        got, rounds = sharded_mst_host(n, lambda: np.stack([e.mst_shard_best(host=True) for e in engs]), lambda k: k, set_components)
        assert rounds > 1 and same(got, want)
    finally:
        for e in engs:
            e.close()


def test_group_call_without_triangles(engine, monkeypatch):
    """lcsgpu_multi_mst_prim with every context in recompute mode (what N GPUs do beyond ~1.5 M sequences)."""
    seqs = _sets()["family"]
    engine.upload_seqs(seqs)
    want = engine.mst_prim(1)
    grp = famsa_amd.LcsGpuGroup([0, 0, 0])
    try:
        grp.upload_seqs(seqs)
        monkeypatch.setenv("LCSGPU_MST_MODE", "recompute")
        assert same(grp.mst_prim(1), want)
    finally:
        grp.close()


def _cli(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([CLI, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    return p


@pytest.mark.parametrize("mode", ["fused", "recompute"])
@pytest.mark.parametrize("gt", ["sl", "slink"])
@pytest.mark.parametrize("case", ["adeno_fiber/adeno_fiber", "hemopexin/hemopexin", "adversarial_tree.fasta"])
def test_cli_goldens_in_both_modes(tmp_path, case, gt, mode):
    gold = {"adeno_fiber/adeno_fiber": f"adeno_fiber/{gt}.dnd", "adversarial_tree.fasta": f"adversarial_tree_{gt}.dnd",
            "hemopexin/hemopexin": f"hemopexin/{gt}.dnd"}[case]
    out = str(tmp_path / "t.dnd")
    _cli("-gt", gt, "-gt_export", os.path.join(G, case), out, env={"LCSGPU_MST_MODE": mode})
    assert open(out, "rb").read() == open(os.path.join(G, gold), "rb").read()


def test_c4_with_part_of_the_triangle(tmp_path):
    """BASELINE config C4 (100 000 x 400 aa) as a set whose triangle does not fit: with 8.5 GB of "free" device memory the
    rows below ~90 000 stay resident (passes) and the rest -- a fifth of the pairs -- is recomputed in each of the 7 rounds
    with the fold fused into the launch; the two blocks' keys merge like two ranks'.  The Newick must be the reference's
    (tests/golden/meta_large.json).  (Nothing resident at all, `recompute`: test_gpu_realmix.py and the smaller sets above;
    at this size it is 7 full LCS passes, 10 s.)"""
    codes, offsets = seqio.synth_uniform(100000, 400)
    path = str(tmp_path / "synth100k.fasta")
    seqio.to_fasta(codes, offsets, path)
    out = str(tmp_path / "sl.dnd")
    p = _cli("-v", "-gt", "sl", "-gt_export", path, out, env={"LCSGPU_FAKE_HBM_GB": "8.5", "LCSGPU_PROFILE": "1"})
    h = hashlib.sha256(open(out, "rb").read()).hexdigest()
    assert h == META_LARGE["synth100k"]["sl_newick_sha256"], p.stderr


def test_the_length_bound_lets_tiles_go_and_changes_nothing(tmp_path):
    """MSTPrim's length bound (reference tree/MSTPrim.cpp:450-467) for whole tiles of the rounds that recompute their LCS
    values (FuseArgs::prune): on a set of very different lengths most far-from-the-diagonal tiles are let go -- `-vv` says how
    many -- and the tree is the one the passes over a resident triangle give, and the one without the bound."""
    rng = np.random.Generator(np.random.PCG64(23))
    fams = []
    for length, members in ((60, 2500), (150, 2500), (420, 2500), (900, 1500)):
        anc = rng.integers(0, 20, size=length, dtype=np.uint8)
        for _ in range(members):
            s = anc.copy()
            m = rng.random(length) < 0.2
            s[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
            fams.append(s[: int(rng.integers(int(length * 0.9), length + 1))].copy())
    order = rng.permutation(len(fams))
    fams = [fams[i] for i in order]
    codes, offsets = seqio.pack(fams)
    path = str(tmp_path / "ragged.fasta")
    seqio.to_fasta(codes, offsets, path)

    def run(env):
        out = str(tmp_path / ("t_%d.dnd" % len(env)))
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([CLI, "-vv", "-gt", "sl", "-gt_export", path, out], stderr=subprocess.PIPE, text=True, env=e, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        return hashlib.sha256(open(out, "rb").read()).hexdigest(), p.stderr

    want, _ = run({"LCSGPU_MST_MODE": "passes"})
    got, log = run({"LCSGPU_MST_MODE": "recompute", "X": "1"})
    assert got == want
    line = [l for l in log.splitlines() if "let go by the length bound" in l]
    assert line, log[-1500:]
    computed, gone = [int(x) for x in __import__("re").findall(r"(\d+) computed, (\d+) let go", line[0])[0]]
    assert gone > computed // 4, line[0]  # the four families are far apart in length: most cross-family tiles go
    off, log_off = run({"LCSGPU_MST_MODE": "recompute", "LCSGPU_TUNE": "mst_length_bound=0", "Y": "22"})
    assert off == want
    assert not any("let go" in l and " 0 let go" not in l for l in log_off.splitlines() if "let go" in l)


def test_kept_records_and_the_multiplication_test_change_no_tree(tmp_path):
    """The passes over a resident triangle take two shortcuts (mst_kernels.hip): a vertex's record of the round before stands
    when its edge still leaves the vertex's component (BoruvkaArgs::keep), and a candidate is proved worse by one
    multiplication before the IEEE division (exact_update_p).  Neither may move an edge: a ragged, tie-heavy set (short
    sequences over four letters: many equal distances, so the id order decides often; lengths 8-70: the integer filter's
    slack is wide) and a family set, each with both shortcuts, with each alone and with none -- and against the rounds that
    recompute, which share neither."""
    rng = np.random.Generator(np.random.PCG64(31))
    seqs = [rng.integers(0, 4, size=int(rng.integers(8, 71)), dtype=np.uint8) for _ in range(3000)]
    anc = rng.integers(0, 20, size=120, dtype=np.uint8)
    for _ in range(1500):
        s = anc.copy()
        m = rng.random(120) < 0.15
        s[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
        seqs.append(s[: int(rng.integers(90, 121))].copy())
    seqs = [seqs[i] for i in rng.permutation(len(seqs))]
    codes, offsets = seqio.pack(seqs)
    path = str(tmp_path / "ties.fasta")
    seqio.to_fasta(codes, offsets, path)

    def run(tag, env, dist="indel075_div_lcs"):
        out = str(tmp_path / ("t_%s.dnd" % tag))
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([CLI, "-gt", "sl", "-dist", dist, "-gt_export", path, out], stderr=subprocess.PIPE, text=True, env=e, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        return hashlib.sha256(open(out, "rb").read()).hexdigest()

    for dist in ("indel075_div_lcs", "indel_div_lcs"):
        want = run("none" + dist, {"LCSGPU_MST_MODE": "passes", "LCSGPU_TUNE": "mst_keep=0,mst_crossmul=0"}, dist)
        assert run("both" + dist, {"LCSGPU_MST_MODE": "passes"}, dist) == want
        assert run("keep" + dist, {"LCSGPU_MST_MODE": "passes", "LCSGPU_TUNE": "mst_crossmul=0"}, dist) == want
        assert run("mul" + dist, {"LCSGPU_MST_MODE": "passes", "LCSGPU_TUNE": "mst_keep=0"}, dist) == want
        assert run("rec" + dist, {"LCSGPU_MST_MODE": "recompute"}, dist) == want
