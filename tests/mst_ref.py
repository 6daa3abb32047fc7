"""Test infrastructure: plain numpy restatements around the sharded single-linkage reduction --
the pair distances from the oracle, MSTPrim's recurrence (reference tree/MSTPrim.cpp:356-533), and the
local half of a Boruvka round over a row block.  Nothing in famsa_amd/ imports this."""
import numpy as np

from famsa_amd.lcsgpu import MST_EDGE, MST_KEY

M64 = (1 << 64) - 1
NO_D = 0x7FEFFFFFFFFFFFFF


def pair_distances(oracle, codes, offsets, kind=1):
    """D[i, j] = D[j, i] = Transform<double>(LCS(ref = max(i,j), partner = min(i,j))): the triangle's orientation."""
    n = len(offsets) - 1
    lens = np.diff(offsets.astype(np.int64))
    tri = oracle.triangle(codes, offsets)
    fn = oracle.lib.oracle_dist_indel075_f64 if kind == 1 else oracle.lib.oracle_dist_indel_f64
    D = np.zeros((n, n), np.float64)
    for i in range(1, n):
        row = tri[i * (i - 1) // 2: i * (i - 1) // 2 + i]
        D[i, :i] = [fn(int(l), int(lens[i]), int(lens[j])) for j, l in enumerate(row)]
        D[:i, i] = D[i, :i]
    return D


def prim_edges(D):
    """MSTPrim::run_view's recurrence over symmetric distances: keys (d, ~pack(min, max)) compared
    lexicographically; returns MST_EDGE[n-1] in insertion order."""
    n = len(D)
    key = [(np.finfo(np.float64).max, 0)] * n
    alive = set(range(1, n))
    cur, out = 0, np.zeros(max(n - 1, 0), MST_EDGE)
    k = 0
    while alive:
        for v in alive:
            d = float(D[cur, v])
            if d <= key[v][0]:
                a, b = min(cur, v), max(cur, v)
                s = (d, M64 ^ ((a << 32) + b))
                if s < key[v]:
                    key[v] = s
        best = min(alive, key=lambda v: key[v])
        packed = M64 ^ key[best][1]
        out[k] = (packed >> 32, packed & 0xFFFFFFFF, key[best][0])
        k += 1
        alive.remove(best)
        cur = best
    return out


def block_best(D, comp, r0, r1):
    """Local half of a Boruvka round over the row block [r0, r1): for every vertex v the smallest key
    (distance bits, ~pack) among the pairs (u, v) the block holds -- (u < v, v in the block) and
    (u > v, u in the block) -- with comp[u] != comp[v]."""
    n = len(D)
    out = np.zeros(n, MST_KEY)
    out["dist_bits"] = NO_D
    out["id"] = M64
    bits = D.view(np.uint64)
    for v in range(n):
        best = (NO_D, M64)
        us = list(range(0, v)) if r0 <= v < r1 else []
        us += list(range(max(v + 1, r0), r1))
        for u in us:
            if comp[u] == comp[v]:
                continue
            a, b = min(u, v), max(u, v)
            k = (int(bits[u, v]), M64 ^ ((a << 32) + b))
            if k < best:
                best = k
        out[v] = best
    return out
