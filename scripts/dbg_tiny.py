import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import famsa_amd
from tests.test_gpu_sharded_mst import _sets, Shards
seqs = _sets()["tiny"]
for parts in (1, 2, 3):
    sh = Shards(seqs, parts)
    for kind in (1, 0, 0x101):
        for flow in ("device", "host", "device"):
            try:
                if flow == "device":
                    r = sh.device_flow(kind)
                    print(parts, kind, flow, "ok rounds", r[1], r[0][0].tolist())
                else:
                    r = sh.host_flow(kind)
                    print(parts, kind, flow, "ok rounds", r[1], r[0].tolist())
            except Exception as e:
                print(parts, kind, flow, "FAILED", e)
    sh.close()
