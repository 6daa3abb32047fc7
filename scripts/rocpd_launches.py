#!/usr/bin/env python3
"""dev tool: the dispatches of a rocprofv3 (rocpd sqlite) kernel trace in start order: start, duration, grid, queue, name.
usage: rocpd_launches.py <results.db> [name-substring] [min-us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
def col(*c):
    return next((x for x in c if x in cols), None)
c_start, c_end, c_name = col("start"), col("end"), col("name", "kernel_name")
c_gx, c_gy, c_wx = col("grid_x", "grid_size_x"), col("grid_y", "grid_size_y"), col("workgroup_x", "workgroup_size_x")
c_q = col("queue_id", "queue", "stream_id", "stream")
sel = [c for c in (c_name, c_start, c_end, c_gx, c_gy, c_wx, c_q) if c]
rows = list(cur.execute(f"select {','.join(sel)} from kernels order by {c_start}"))
t0 = rows[0][1]
for r in rows:
    us = (r[2] - r[1]) / 1e3
    if pat in r[0] and us >= min_us:
        print(f"{(r[1] - t0) / 1e6:9.3f} ms  {us:9.1f} us  grid {r[3]}x{r[4]} wg {r[5]}  q {r[6]}  {r[0][:70]}")
