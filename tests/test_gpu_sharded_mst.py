"""Multi-GPU single linkage on ONE GPU: 1, 2 and 3 engine contexts on cuda:0, each holding a disjoint row
block of the LCS triangle (what the GPUs of a node hold), Boruvka rounds with the per-round key exchange
done (a) in device memory -- the form bench.py runs over RCCL -- and (b) in host memory with the library's
host merge -- the form famsa-gpu -gpus N runs.  Edges, their order and distances must be bit-identical to the
single-context lcsgpu_mst_prim, which the oracle's recurrence pins (test_gpu_parity.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import famsa_amd
from famsa_amd import seqio
from famsa_amd.lcsgpu import MST_TRIANGLE_ORIENTATION
from famsa_amd.rowblock import row_cuts, pairs_in_rows, sharded_mst_host, edge_list_sha256

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _same(a, b):
    return (a["from"] == b["from"]).all() and (a["to"] == b["to"]).all() and \
        (a["dist"].view(np.uint64) == b["dist"].view(np.uint64)).all()


def _sets():
    rng = np.random.Generator(np.random.PCG64(53))
    ties = [rng.integers(0, 3, size=int(rng.integers(3, 12))).astype(np.uint8) for _ in range(1500)]
    anc = rng.integers(0, 20, size=200, dtype=np.uint8)
    fam = []
    for _ in range(3000):
        s = anc.copy()
        m = rng.random(200) < 0.2
        s[m] = rng.integers(0, 20, size=int(m.sum()), dtype=np.uint8)
        fam.append(s[: int(rng.integers(120, 201))].copy())
    uni = [rng.integers(0, 20, size=400).astype(np.uint8) for _ in range(2500)]
    tiny = [rng.integers(0, 20, size=int(rng.integers(5, 40))).astype(np.uint8) for _ in range(3)]
    return {"ties": ties, "family": fam, "uniform": uni, "tiny": tiny}


class Shards:
    """`parts` contexts on cuda:0 over the same uploaded set, context p holding rows [cuts[p], cuts[p+1])."""

    def __init__(self, seqs, parts):
        import torch
        self.torch = torch
        self.n = len(seqs)
        self.parts = parts
        self.cuts = row_cuts(self.n, parts)
        self.engs = [famsa_amd.LcsGpu(0) for _ in range(parts)]
        self.tris = []
        for p, e in enumerate(self.engs):
            e.upload_seqs(seqs)
            r0, r1 = self.cuts[p], self.cuts[p + 1]
            t = torch.empty(max(pairs_in_rows(r0, r1), 1), dtype=torch.int16, device="cuda:0")
            e.lcs_triangle_dev(r0, r1, t.data_ptr(), 2, sync=True)
            self.tris.append(t)

    def begin(self, kind):
        for p, e in enumerate(self.engs):
            e.mst_shard_begin(self.tris[p].data_ptr(), 2, self.cuts[p], self.cuts[p + 1], kind)

    def device_flow(self, kind):
        torch = self.torch
        self.begin(kind)
        keys = [torch.zeros(2 * self.n, dtype=torch.int64, device="cuda:0") for _ in self.engs]
        torch.cuda.synchronize()  # torch fills them on ITS stream; the engines write them on theirs
        found, rounds = 0, 0
        while found < self.n - 1:
            assert rounds < 40
            for e, k in zip(self.engs, keys):
                e.mst_shard_best(k.data_ptr())
            for e in self.engs:
                e.sync()
            gathered = torch.cat(keys)          # the all-gather: every context sees every block's keys
            torch.cuda.synchronize()
            counts = [e.mst_shard_merge(gathered.data_ptr(), self.parts) for e in self.engs]
            assert len(set(counts)) == 1        # replicated state: every context found the same edges
            found = counts[0]
            rounds += 1
        return [e.mst_shard_finish() for e in self.engs], rounds

    def host_flow(self, kind):
        self.begin(kind)

        def local_best():
            return np.stack([e.mst_shard_best(host=True) for e in self.engs])

        def set_components(comp):
            for e in self.engs:
                e.mst_shard_set_components(comp)

        return sharded_mst_host(self.n, local_best, lambda k: k, set_components)

    def close(self):
        for e in self.engs:
            e.close()


@pytest.mark.parametrize("name", ["ties", "family", "uniform", "tiny"])
def test_row_block_contexts_end_in_the_single_context_tree(engine, name):
    seqs = _sets()[name]
    engine.upload_seqs(seqs)
    assert engine.orientation_flags().sum() == 0
    for parts in (1, 2, 3):
        sh = Shards(seqs, parts)
        try:
            for kind in (1, 0, 1 | MST_TRIANGLE_ORIENTATION):
                want = engine.mst_prim(kind)
                per_ctx, rounds = sh.device_flow(kind)
                for got in per_ctx:
                    assert _same(got, want), (name, parts, kind)
                got, rounds_h = sh.host_flow(kind)
                assert _same(got, want), (name, parts, kind, "host merge")
                assert rounds == rounds_h
        finally:
            sh.close()


def test_small_set_against_the_recurrence(engine, oracle):
    """The whole chain against MSTPrim's recurrence in plain Python over oracle LCS (not only against the
    engine's own single-context path), more contexts than some blocks have rows."""
    import mst_ref
    rng = np.random.Generator(np.random.PCG64(8))
    seqs = [rng.integers(0, 4, size=int(rng.integers(4, 30))).astype(np.uint8) for _ in range(150)]
    codes, offsets = seqio.pack(seqs)
    for kind in (1, 0):
        want = mst_ref.prim_edges(mst_ref.pair_distances(oracle, codes, offsets, kind))
        for parts in (2, 5):
            sh = Shards(seqs, parts)
            try:
                per_ctx, _ = sh.device_flow(kind)
                assert all(_same(g, want) for g in per_ctx)
                assert _same(sh.host_flow(kind)[0], want)
            finally:
                sh.close()


def test_orientation_sensitive_sets(engine):
    """With a carry-quirk sequence MSTPrim's own orientation is refused (the triangle does not hold those
    distances; lcsgpu_mst_prim keeps its step-by-step kernel); SLINK's orientation works on row blocks."""
    ids, res = seqio.read_fasta(os.path.join(G, "adversarial_tree.fasta"))
    enc = [famsa_amd.lcsgpu.encode(r) for r in res]
    enc = [enc[i] for i in seqio.sort_order(enc)]
    rng = np.random.Generator(np.random.PCG64(3))
    enc += [rng.integers(0, 20, size=int(l)).astype(np.uint8) for l in rng.integers(40, 300, size=400)]
    engine.upload_seqs(enc)
    assert engine.orientation_flags().sum() >= 2
    sh = Shards(enc, 3)
    try:
        with pytest.raises(famsa_amd.LcsGpuError, match="orientation sensitive"):
            sh.begin(1)
        want = engine.mst_prim(1 | MST_TRIANGLE_ORIENTATION)
        per_ctx, _ = sh.device_flow(1 | MST_TRIANGLE_ORIENTATION)
        assert all(_same(g, want) for g in per_ctx)
        assert _same(sh.host_flow(1 | MST_TRIANGLE_ORIENTATION)[0], want)
    finally:
        sh.close()


def test_row_minima_of_a_row_block(engine, oracle):
    """row_minima_kernel with row_begin > 0 (a rank's block), against the oracle."""
    import torch
    rng = np.random.Generator(np.random.PCG64(21))
    seqs = [rng.integers(0, 6, size=int(rng.integers(20, 90))).astype(np.uint8) for _ in range(700)]
    engine.upload_seqs(seqs)
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    lens = np.diff(offsets.astype(np.int64))
    lcs = oracle.triangle(codes, offsets)
    cuts = row_cuts(n, 3)
    for kind, fn in [(1, oracle.lib.oracle_dist_indel075_f64), (0, oracle.lib.oracle_dist_indel_f64)]:
        for r0, r1 in zip(cuts, cuts[1:]):
            tri = torch.empty(max(pairs_in_rows(r0, r1), 1), dtype=torch.int16, device="cuda:0")
            engine.lcs_triangle_dev(r0, r1, tri.data_ptr(), 2)
            out = torch.zeros((r1 - r0, 2), dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()  # torch fills on ITS stream; the engine writes `out` on its own
            engine.row_minima_dev(tri.data_ptr(), 2, r0, r1, kind, out.data_ptr(), sync=True)
            d = out[:, 0].cpu().numpy()
            j = out[:, 1].cpu().numpy().view(np.int64)
            for i in list(range(max(r0, 1), min(r0 + 25, r1))) + list(range(r0 + 25, r1, 23)) + [r1 - 1]:
                if i < 1:
                    continue
                row = lcs[i * (i - 1) // 2: i * (i - 1) // 2 + i]
                dd = np.array([fn(int(l), int(lens[i]), int(lens[k])) for k, l in enumerate(row)])
                m = dd.min()
                assert d[i - r0] == m and j[i - r0] == int(np.max(np.nonzero(dd == m)[0])), (kind, r0, i)


_ONE_RANK = {}


def _bench(args, world, torchrun=True):
    if world == 1 and tuple(args) in _ONE_RANK:  # several tests compare with the same one-rank line
        return _ONE_RANK[tuple(args)]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if world == 1 or not torchrun:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(29500 + world), os.path.join(ROOT, "bench.py"),
               "--emulate-ranks-on-one-gpu"]
    p = subprocess.run(cmd + ["--gpus", str(world)] + args, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    if world == 1:
        _ONE_RANK[tuple(args)] = rec
    return rec


def test_bench_step_ends_in_the_same_tree_for_every_rank_count():
    """bench.py --gpus N (ranks sharing cuda:0, gloo exchange): the MST edge-list hash it prints does not
    depend on N, and the sampled oracle check of the timed triangle passes inside it."""
    args = ["--steps", "1", "--warmup", "1", "--n-seqs", "5000", "--seq-len", "120", "--no-cpu-baseline"]
    one = _bench(args, 1)
    two = _bench(args, 2)  # under torch.distributed.run, as the driver starts it
    assert one["mst"]["edges_sha256"] == two["mst"]["edges_sha256"]
    assert one["mst"]["n_edges"] == 4999 and two["n_gpus"] == 2
    assert one["parity"]["sampled_pairs"] > 0 and one["parity"]["mismatches"] == 0
    assert two["parity"]["mismatches"] == 0


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` WITHOUT torchrun (how the N = 1 record is produced, and what a driver that starts
    N > 1 the same way gets): the script starts its N ranks itself, runs the fail-closed self-check (the ranks' sharded
    tree on a 2000-sequence family set against every rank's single-context tree) and prints one line with the per-rank
    kernel times; the hash equals the one-rank hash."""
    args = ["--steps", "1", "--warmup", "1", "--n-seqs", "5000", "--seq-len", "120", "--no-cpu-baseline"]
    one = _bench(args, 1)
    three = _bench(args + ["--emulate-ranks-on-one-gpu"], 3, torchrun=False)
    for rec, w in ((three, 3),):
        assert rec["n_gpus"] == w and rec["mst"]["edges_sha256"] == one["mst"]["edges_sha256"]
        r = rec["ranks"]
        assert r["world"] == w and len(r["kernel_ms_per_rank"]) == w and r["kernel_ms_min"] > 0
        assert r["self_check"]["ranks_agree"] and r["self_check"]["n_seqs"] == 2000
        assert rec["parity"]["mismatches"] == 0
        # what the ranks talk over, collected before anything was timed
        t = r["transport_report"]
        assert t["devices_visible"] >= 1 and isinstance(t["peer_access"], list) and "rccl_version" in t
    assert one["ranks"]["self_check"] is None and one["ranks"]["rccl_ranks"] is None


def test_bench_contexts_mode():
    """--mode contexts: one process, N contexts, lcsgpu_multi_mst_prim -- the library's own multi-GPU path -- with its
    self-check; same tree as the ranks mode, per-context kernel times, the transport report in the line."""
    args = ["--steps", "1", "--warmup", "1", "--n-seqs", "5000", "--seq-len", "120", "--no-cpu-baseline"]
    one = _bench(args, 1)
    ctx3 = _bench(args + ["--mode", "contexts", "--emulate-ranks-on-one-gpu"], 3, torchrun=False)
    assert ctx3["mst"]["edges_sha256"] == one["mst"]["edges_sha256"] and ctx3["n_gpus"] == 3
    r = ctx3["ranks"]
    assert r["mode"] == "contexts" and len(r["kernel_ms_per_rank"]) == 3 and r["kernel_ms_min"] > 0
    assert r["self_check"]["contexts_agree"] and "peer copies" in r["transport"]
    assert ctx3["parity"]["sampled_pairs"] > 0 and ctx3["parity"]["mismatches"] == 0


def test_bench_with_the_real_collective_on_one_rank():
    """bench.py --force-collective: one rank, but the exchange of every Boruvka round is the N > 1 path's own --
    torch.distributed initialised with the nccl (= RCCL) backend, all_gather_into_tensor on device tensors, the
    hand-off between the engine's stream and torch's -- so RCCL initialisation and the stream ordering run on an
    MI355X in every GPU test run, not for the first time on an 8-GPU node.  Same tree as without it."""
    args = ["--steps", "2", "--warmup", "1", "--n-seqs", "6000", "--seq-len", "120", "--no-cpu-baseline"]
    plain = _bench(args, 1)
    forced = _bench(args + ["--force-collective", "--self-check"], 1)
    assert forced["config"]["exchange"].startswith("nccl") and plain["config"]["exchange"].startswith("none")
    assert forced["ranks"]["rccl_ranks"] == 1 and forced["ranks"]["self_check"]["ranks_agree"]
    assert forced["mst"]["edges_sha256"] == plain["mst"]["edges_sha256"]
    assert forced["mst"]["rounds"] == plain["mst"]["rounds"] and forced["parity"]["mismatches"] == 0
