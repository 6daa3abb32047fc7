"""Batched leaf reducers on the device (include/lcsgpu.h, lcsgpu_leaf_upgma_batch): the <= 2000-sequence UPGMA
sub-trees of the FastTree recursion, one workgroup per leaf, all leaves of a split in one launch.  Against the host
restatement of UPGMA::computeTree over the oracle's LCS values (itself pinned to the reference by the CPU suite), on
the shapes that decide ties; and end to end against the upstream medoid goldens / the at-size reference values with
the leaves on the device (FAMSA_LEAF_DEVICE=1) and on the host (the default: at 3 000 000 sequences the device form
measured slower end to end -- long-running leaf workgroups slow the CLARANS rounds they share the chip with)."""
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
G = os.path.join(ROOT, "tests", "golden")
META_LARGE = json.load(open(os.path.join(G, "meta_large.json")))


def host_upgma(oracle, seqs, modified, kind):
    """UPGMA::computeTree restated in numpy float32 over oracle distances (reference tree/UPGMA.cpp:114-295)."""
    m = len(seqs)
    codes, offsets = seqio.pack(seqs)
    lens = np.array([len(s) for s in seqs], np.uint32)
    D = oracle.dist_triangle_f32(oracle.triangle(codes, offsets), lens, kind=kind)

    def at(i, j):
        return j + i * (i - 1) // 2 if i >= j else i + j * (j - 1) // 2
    BIG = np.float32(1e29)
    mind = np.full(m, BIG, np.float32)
    near = np.full(m, 0x7FFFFFFF, np.int64)
    node = np.arange(m, dtype=np.int64)
    for i in range(1, m):
        for j in range(i):
            d = D[at(i, j)]
            if d < mind[i]:
                mind[i], near[i] = d, j
            if d < mind[j]:
                mind[j], near[j] = d, i
    left, right = [], []
    for it in range(m - 1):
        L, best = -1, BIG
        for j in range(m):
            if node[j] != 0x7FFFFFFF and mind[j] < best:
                best, L = mind[j], j
        R = int(near[L])
        nd, nj = BIG, 0x7FFFFFFF
        for j in range(m):
            if j == L or j == R or node[j] == 0x7FFFFFFF:
                continue
            x, y = D[at(L, j)], D[at(R, j)]
            if modified:
                v = np.float32(np.float32(0.05) * np.float32(x + y)) + np.float32(np.float32(0.9) * min(x, y))
            else:
                v = np.float32(np.float32(x + y) * np.float32(0.5))
            v = np.float32(v)
            if near[j] == R:
                near[j] = L
            D[at(L, j)] = v
            if v < nd:
                nd, nj = v, j
        left.append(int(node[L]))
        right.append(int(node[R]))
        node[L] = m + it
        near[L], mind[L] = nj, nd
        node[R] = 0x7FFFFFFF
    return np.array(left, np.int32), np.array(right, np.int32)


def test_leaf_batch_against_the_host_algorithm(engine, oracle):
    rng = np.random.Generator(np.random.PCG64(77))
    seqs = [rng.integers(0, 3, size=int(rng.integers(6, 16))).astype(np.uint8) for _ in range(700)]   # ties everywhere
    seqs += seqio.synth_family(600, 120, seed=4)
    engine.upload_seqs(seqs)
    perm = rng.permutation(len(seqs))
    # lists of 1, 2, 3, 65, 257 and ~300 members, mixed tie-heavy / family members, unsorted ids
    groups = [perm[:1], perm[1:3], perm[3:6], perm[6:71], perm[71:328], perm[328:640], np.arange(700, 1000)[::-1]]
    for kind in (1, 0):
        for modified in (False, True):
            got = engine.leaf_upgma_batch(groups, kind=kind, modified=modified)
            for g, (l, r) in zip(groups, got):
                wl, wr = host_upgma(oracle, [seqs[int(i)] for i in g], modified, kind)
                assert (l == wl).all() and (r == wr).all(), (len(g), kind, modified)


def test_leaf_batch_limits(engine):
    seqs = seqio.synth_family(2100, 60, seed=8)
    engine.upload_seqs(seqs)
    with pytest.raises(famsa_amd.LcsGpuError, match="members"):
        engine.leaf_upgma_batch([np.arange(2100)])
    one = engine.leaf_upgma_batch([np.arange(2048)])[0]  # the largest list it takes: 2047 merges in one workgroup
    assert len(one[0]) == 2047 and one[0].max() <= 2 * 2048 - 3


def _cli(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([CLI, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    return p


@pytest.mark.parametrize("where", ["device", "host"])
def test_medoid_goldens_with_leaves_on_either_side(tmp_path, where):
    env = {"FAMSA_LEAF_DEVICE": "1"} if where == "device" else {}
    f = os.path.join(G, "hemopexin", "hemopexin")
    out = str(tmp_path / "t.dnd")
    _cli("-medoidtree", "-gt", "upgma", "-gt_export", f, out, env=env)
    assert open(out, "rb").read() == open(os.path.join(G, "hemopexin", "medoid-upgma.dnd"), "rb").read()
    _cli("-medoidtree", "-gt", "upgma", "-gt_export", "-subtree_size", "10", "-sample_size", "100", "-medoid_threshold", "100",
         "-cluster_fraction", "0.2", "-cluster_iters", "1", f, out, env=env)
    assert open(out, "rb").read() == open(os.path.join(G, "hemopexin", "medoid-upgma-params.dnd"), "rb").read()


def test_family_200k_with_leaves_on_the_device(tmp_path):
    """(The default -- leaves on the host -- is pinned at 200 000, 1 000 000 and 3 000 000 sequences by
    test_gpu_atsize.py; this covers the device form at size: thousands of leaves of every size up to the threshold.)"""
    rec = META_LARGE["family200000"]
    path = str(tmp_path / "family.fasta")
    seqio.family_fasta(200000, rec["len"], path)
    out = str(tmp_path / "medoid.dnd")
    _cli("-medoidtree", "-gt", "upgma", "-gt_export", path, out, env={"FAMSA_LEAF_DEVICE": "1"})
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == rec["medoid_upgma_newick_sha256"]
