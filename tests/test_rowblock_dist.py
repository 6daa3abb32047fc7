"""Multi-process CPU test (gloo, world_size 2) of the N > 1 path: row-block split of the triangle,
per-rank row minima, all-gather, assembly -- against the single-process result.  The per-row minima
are produced by the oracle here (no GPU in this test); on GPUs the same code runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _row_minima(oracle, lcs_tri, lens, r0, r1):
    """(dist, index) of rows [r0, r1): MSTPrim's key order -- smaller distance, then larger column."""
    out = np.zeros((r1 - r0, 2), np.float64)
    idx = out[:, 1].view(np.int64)
    for i in range(r0, r1):
        if i == 0:
            out[0, 0] = np.finfo(np.float64).max
            idx[0] = -1
            continue
        row = lcs_tri[i * (i - 1) // 2: i * (i - 1) // 2 + i]
        d = np.array([oracle.lib.oracle_dist_indel075_f64(int(l), int(lens[i]), int(lens[j])) for j, l in enumerate(row)])
        m = d.min()
        out[i - r0, 0] = m
        idx[i - r0] = int(np.max(np.nonzero(d == m)[0]))
    return out


def _worker(rank, world, port, n, q):
    import oracle_bind
    from famsa_amd import seqio
    from famsa_amd.rowblock import row_cuts, max_block_rows, allgather_row_minima, assemble_row_minima
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = oracle_bind.Oracle()
    codes, offsets = seqio.synth_uniform(n, 60, seed=99)
    lens = np.diff(offsets.astype(np.int64))
    tri = oracle.triangle(codes, offsets)
    cuts = row_cuts(n, world)
    r0, r1 = cuts[rank], cuts[rank + 1]
    local = torch.zeros(max_block_rows(cuts) * 2, dtype=torch.float64)
    rec = _row_minima(oracle, tri, lens, r0, r1)
    local[: rec.size] = torch.from_numpy(rec.reshape(-1))
    gathered = allgather_row_minima(local, cuts)
    d, j = assemble_row_minima(gathered, cuts)
    if rank == 0:
        full = _row_minima(oracle, tri, lens, 0, n)
        ok = bool((d.numpy() == full[:, 0]).all() and (j.numpy() == full[:, 1].view(np.int64)).all())
        q.put((ok, cuts))
    dist.barrier()
    dist.destroy_process_group()


def test_row_cuts_balance():
    from famsa_amd.rowblock import row_cuts, pairs_in_rows
    for n, parts in [(100000, 8), (10000, 4), (1001, 2), (5, 8), (0, 2), (1, 1)]:
        cuts = row_cuts(n, parts)
        assert cuts[0] == 0 and cuts[-1] == n and all(a <= b for a, b in zip(cuts, cuts[1:]))
        total = sum(pairs_in_rows(a, b) for a, b in zip(cuts, cuts[1:]))
        assert total == n * (n - 1) // 2 if n > 0 else total == 0
    cuts = row_cuts(100000, 8)
    per = [pairs_in_rows(a, b) for a, b in zip(cuts, cuts[1:])]
    assert max(per) / min(per) < 1.001


def test_two_rank_allgather_of_row_minima():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 300, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, cuts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok, cuts
