"""The multi-GPU single-linkage reduction (Boruvka over row blocks, include/lcsgpu.h lcsgpu_mst_shard_* /
lcsgpu_mst_merge_host), CPU side: the global half of a round (pure host code of the library), the round
driver, and the exchange over torch.distributed (gloo, world_size 2) -- against MSTPrim's recurrence
(reference tree/MSTPrim.cpp:356-533) over oracle distances.  The local half is produced by a numpy
restatement here; on GPUs it is lcsgpu_mst_shard_best (tests/test_gpu_sharded_mst.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _sets():
    from famsa_amd import seqio
    rng = np.random.Generator(np.random.PCG64(77))
    ties = [rng.integers(0, 3, size=int(rng.integers(3, 10))).astype(np.uint8) for _ in range(90)]
    fam = seqio.synth_family(120, 60, seed=5)
    uni = [rng.integers(0, 20, size=50).astype(np.uint8) for _ in range(100)]
    return {"ties": ties, "family": fam, "uniform": uni}


def _same(a, b):
    return (a["from"] == b["from"]).all() and (a["to"] == b["to"]).all() and \
        (a["dist"].view(np.uint64) == b["dist"].view(np.uint64)).all()


@pytest.mark.parametrize("name", ["ties", "family", "uniform"])
@pytest.mark.parametrize("parts", [1, 2, 3, 7])
def test_rounds_over_row_blocks_equal_prim(oracle, name, parts):
    import mst_ref
    from famsa_amd import seqio
    from famsa_amd.rowblock import row_cuts, sharded_mst_host
    seqs = _sets()[name]
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    for kind in (1, 0):
        D = mst_ref.pair_distances(oracle, codes, offsets, kind)
        want = mst_ref.prim_edges(D)
        cuts = row_cuts(n, parts)
        state = {"comp": np.arange(n, dtype=np.int32)}

        def local_best():
            return np.stack([mst_ref.block_best(D, state["comp"], cuts[p], cuts[p + 1]) for p in range(parts)])

        got, rounds = sharded_mst_host(n, local_best, lambda k: k, lambda c: state.update(comp=c.copy()))
        assert _same(got, want), (name, parts, kind)
        assert rounds <= int(np.ceil(np.log2(n))) + 1


def test_merge_host_rejects_bad_input():
    from famsa_amd.lcsgpu import MST_EDGE, MST_KEY, LcsGpuError, mst_merge_host, mst_order_edges
    keys = np.zeros((1, 4), MST_KEY)
    keys["dist_bits"] = 0x7FEFFFFFFFFFFFFF
    keys["id"] = np.uint64(0xFFFFFFFFFFFFFFFF)
    comp = np.arange(4, dtype=np.int32)
    edges = np.zeros(3, MST_EDGE)
    assert mst_merge_host(keys, comp, edges, 0) == 0          # no candidates: nothing happens
    comp[2] = 9
    keys[0, 2] = (0, np.uint64(0xFFFFFFFFFFFFFFFF) ^ np.uint64((1 << 32) + 2))
    with pytest.raises(LcsGpuError):
        mst_merge_host(keys, comp, edges, 0)                    # label out of range
    bad = np.zeros(3, MST_EDGE)
    bad["from"], bad["to"] = [0, 0, 1], [1, 1, 0]
    with pytest.raises(LcsGpuError):
        mst_order_edges(bad, 4)                                 # from >= to
    cyc = np.zeros(3, MST_EDGE)
    cyc["from"], cyc["to"] = [0, 0, 1], [1, 2, 2]
    with pytest.raises(LcsGpuError):
        mst_order_edges(cyc, 4)                                 # does not span vertex 3


def _worker(rank, world, port, q):
    import mst_ref
    import oracle_bind
    from famsa_amd import seqio
    from famsa_amd.rowblock import row_cuts, sharded_mst_host, allgather_keys_host, edge_list_sha256
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = oracle_bind.Oracle()
    seqs = _sets()["ties"] + _sets()["family"]
    codes, offsets = seqio.pack(seqs)
    n = len(seqs)
    D = mst_ref.pair_distances(oracle, codes, offsets, 1)
    cuts = row_cuts(n, world)
    state = {"comp": np.arange(n, dtype=np.int32)}
    # every rank: its own block's keys -> all-gather over gloo -> the library's host merge -> same labels everywhere
    got, rounds = sharded_mst_host(n, lambda: mst_ref.block_best(D, state["comp"], cuts[rank], cuts[rank + 1]),
                                   allgather_keys_host, lambda c: state.update(comp=c.copy()))
    q.put((rank, edge_list_sha256(got), rounds, edge_list_sha256(mst_ref.prim_edges(D)) if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_gloo_end_in_the_same_tree():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == res[0][3], res   # both ranks: the tree of the single-process Prim recurrence
