"""Row-block sharding of the all-pairs triangle across the GPUs of one node.

Rows of the lower triangle are independent (row i = ref i against partners j < i), so rank r of N
computes rows [cuts[r], cuts[r+1]) with no data-path collective.  The only exchange is the one the
single-linkage consumer needs: Boruvka rounds over the row-block-resident triangles
(include/lcsgpu.h, lcsgpu_mst_shard_*), one all-gather of n x 16 bytes per rank and round (every
vertex's best edge into another component among the rank's pairs; round 0 = the per-row minima
completed by the per-column minima) -- RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU
tests.  The component labels are a pure function of the gathered keys, so they stay replicated
without a second collective; <= log2(n) rounds (6 at n = 100 000 on the uniform set).
"""
import math

import torch
import torch.distributed as dist


def row_cuts(n, parts):
    """Row-block boundaries with (nearly) equal PAIR counts: row i holds i pairs, so the k-th cut
    sits at n*sqrt(k/parts)."""
    cuts = [int(round(n * math.sqrt(k / parts))) for k in range(parts + 1)]
    cuts[0], cuts[-1] = 0, n
    for k in range(1, parts + 1):
        cuts[k] = max(cuts[k], cuts[k - 1])
    return cuts


def pairs_in_rows(r0, r1):
    return r1 * (r1 - 1) // 2 - r0 * (r0 - 1) // 2


def max_block_rows(cuts):
    return max(cuts[k + 1] - cuts[k] for k in range(len(cuts) - 1))


def allgather_row_minima(local, cuts, gathered=None, group=None):
    """local: float64 tensor [max_block_rows*2] holding this rank's (dist, index-as-bits) records in its
    first (cuts[r+1]-cuts[r]) rows.  Returns (gathered buffer [world*max_rows*2]); use
    `assemble_row_minima` to view it as n records."""
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local, group=group)
    return gathered


def assemble_row_minima(gathered, cuts):
    """[world*max_rows*2] -> (dist[n] float64, index[n] int64) in row order."""
    world = len(cuts) - 1
    max_rows = max_block_rows(cuts)
    g = gathered.view(world, max_rows, 2)
    parts = [g[r, : cuts[r + 1] - cuts[r]] for r in range(world)]
    allrec = torch.cat(parts, dim=0)
    return allrec[:, 0].contiguous(), allrec[:, 1].contiguous().view(torch.int64)


# ---- the sharded single-linkage reduction: Boruvka rounds with one all-gather per round ----------

def sharded_mst_device(eng, d_tri_ptr, elem_size, r0, r1, kind, keys, gathered, all_gather, max_rounds=64, ordered=True,
                       begun=False):
    """Device flow (bench.py over RCCL; the multi-context GPU test): `eng` holds rows [r0, r1) at
    d_tri_ptr.  keys: device int64 tensor [2n] (this rank's lcsgpu_mst_key records), gathered: device
    int64 tensor [world * 2n]; all_gather(gathered, keys) must order itself after the engine's stream
    and the next engine call after it.  begun: the caller has already called mst_shard_begin (e.g. with
    MST_COMPUTE, which fills d_tri_ptr and does round 0's local half in the same launch).
    Returns (edges in Prim's insertion order, rounds)."""
    n = eng.n
    world = gathered.numel() // max(keys.numel(), 1)
    if not begun:
        eng.mst_shard_begin(d_tri_ptr, elem_size, r0, r1, kind)
    found, rounds = 0, 0
    while found < n - 1:
        if rounds >= max_rounds:
            raise RuntimeError("sharded MST: no convergence")
        eng.mst_shard_best(keys.data_ptr())
        all_gather(gathered, keys)
        found = eng.mst_shard_merge(gathered.data_ptr(), world)
        rounds += 1
    return eng.mst_shard_finish(ordered), rounds


def sharded_mst_host(n, local_best, exchange, set_components, max_rounds=64):
    """Host flow (exchange in host memory: gloo / threads of one process): local_best() -> this part's
    MST_KEY[n] (lcsgpu_mst_shard_best with a host buffer), exchange(keys) -> MST_KEY[parts, n] (every
    part's keys), set_components(comp int32[n]) hands the new labels to the part(s).  The global half of
    each round runs in lcsgpu_mst_merge_host (pure host code).  Returns (edges in Prim's order, rounds)."""
    import numpy as np
    from .lcsgpu import MST_EDGE, mst_merge_host, mst_order_edges
    comp = np.arange(n, dtype=np.int32)
    edges = np.zeros(max(n - 1, 0), dtype=MST_EDGE)
    found, rounds = 0, 0
    while found < n - 1:
        if rounds >= max_rounds:
            raise RuntimeError("sharded MST: no convergence")
        allk = exchange(local_best())
        before = found
        found = mst_merge_host(allk, comp, edges, found)
        if found <= before:
            raise RuntimeError("sharded MST: a round added no edge")
        set_components(comp)
        rounds += 1
    return mst_order_edges(edges, n), rounds


def allgather_keys_host(keys, group=None):
    """MST_KEY[n] of this rank -> MST_KEY[world, n] through torch.distributed (any backend with CPU tensors)."""
    import numpy as np
    from .lcsgpu import MST_KEY
    world = dist.get_world_size(group)
    local = torch.from_numpy(np.ascontiguousarray(keys).view(np.int64).reshape(-1))
    out = torch.empty(world * local.numel(), dtype=torch.int64)
    dist.all_gather_into_tensor(out, local, group=group)
    return out.numpy().view(MST_KEY).reshape(world, len(keys))


def edge_list_sha256(edges):
    """Hash of the ordered edge list (from, to, distance bits): equal for every number of ranks."""
    import hashlib
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(edges).tobytes()).hexdigest()
