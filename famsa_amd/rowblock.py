"""Row-block sharding of the all-pairs triangle across the GPUs of one node.

Rows of the lower triangle are independent (row i = ref i against partners j < i), so rank r of N
computes rows [cuts[r], cuts[r+1]) with no data-path collective.  The only exchange is the one the
single-linkage consumer needs: every rank's per-row minima (16 bytes per row: distance f64 + column
i64), all-gathered so that each rank holds all n records -- RCCL over xGMI on GPUs
(backend "nccl"), gloo in the CPU tests.
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
