"""Round 6: what the Boruvka rounds of the 13 774-record set look like -- components, the largest one, the vertices outside it --
to judge what a round that only looks at the rows and columns of those vertices would save.
Run on a GPU box: python scripts/mst_rounds_stats.py"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import famsa_amd  # noqa: E402
from famsa_amd.lcsgpu import MST_EDGE, mst_merge_host  # noqa: E402

args = types.SimpleNamespace(workload="realmix", order="sorted", n=0, len=0)
codes, offsets, what, _ = bench.make_workload(args)
n = len(offsets) - 1
print(what)
import torch  # noqa: E402

eng = famsa_amd.LcsGpu(0)
eng.upload(codes, offsets)
tri = torch.empty(n * (n - 1) // 2, dtype=torch.int16, device="cuda:0")
torch.cuda.synchronize()
eng.lcs_triangle_dev(0, n, tri.data_ptr(), 2)
eng.mst_shard_begin(tri.data_ptr(), 2, 0, n, 1)
comp = np.arange(n, dtype=np.int32)
edges = np.zeros(n - 1, dtype=MST_EDGE)
found, rnd = 0, 0
while found < n - 1:
    keys = eng.mst_shard_best(host=True)[None, :]
    found = mst_merge_host(keys, comp, edges, found)
    eng.mst_shard_set_components(comp)
    labels, counts = np.unique(comp, return_counts=True)
    big = counts.max()
    outside = n - big
    full = n * (n - 1) // 2
    sparse = outside * n
    print(f"after round {rnd}: {len(labels)} components, the largest {big} ({100.0 * big / n:.1f} %), {outside} vertices outside it; "
          f"their rows and columns hold {sparse / full * 100:.1f} % of the triangle's pairs")
    rnd += 1
