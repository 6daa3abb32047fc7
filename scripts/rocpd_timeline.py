#!/usr/bin/env python3
"""dev tool: per-dispatch view of a rocprofv3 (rocpd sqlite) kernel trace: duration percentiles per kernel, by grid
size, and how many dispatches are in flight together.  usage: rocpd_timeline.py <results.db> [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "clarans"
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in names else None
if not view:
    print("views:", names)
    sys.exit(0)
cols = [d[0] for d in cur.execute(f"select * from {view} limit 1").description]
print("columns:", cols)
def col(*cands):
    for c in cands:
        if c in cols:
            return c
    return None
c_start, c_end, c_name = col("start"), col("end"), col("name", "kernel_name")
c_gy, c_gx = col("grid_y", "grid_size_y"), col("grid_x", "grid_size_x")
c_q = col("queue_id", "queue", "stream_id", "stream")
sel = [c_name, c_start, c_end] + [c for c in (c_gx, c_gy, c_q) if c]
rows = list(cur.execute(f"select {','.join(sel)} from {view} order by {c_start}"))
print(len(rows), "dispatches")
import collections
by = collections.defaultdict(list)
for r in rows:
    if pat in r[0]:
        by[r[0][:60]].append(r)
for k, v in by.items():
    d = sorted((r[2] - r[1]) / 1e3 for r in v)
    q = lambda f: d[min(len(d) - 1, int(f * len(d)))]
    print(f"{k}: n={len(d)} p10={q(.1):.1f} p50={q(.5):.1f} p90={q(.9):.1f} p99={q(.99):.1f} max={d[-1]:.1f} us")
    if c_gy:
        i = sel.index(c_gy)
        g = collections.defaultdict(list)
        for r in v:
            g[r[i]].append((r[2] - r[1]) / 1e3)
        print("   by grid_y:", " ".join(f"{gy}:{sum(x)/len(x):.1f}us(n={len(x)})" for gy, x in sorted(g.items())))
    if c_q:
        i = sel.index(c_q)
        print("   queues/streams:", collections.Counter(r[i] for r in v).most_common(8))
# concurrency: time-weighted number of matching dispatches in flight
ev = []
for r in rows:
    if pat in r[0]:
        ev.append((r[1], 1)); ev.append((r[2], -1))
ev.sort()
cur_n, last, hist = 0, None, collections.Counter()
for t, dlt in ev:
    if last is not None:
        hist[cur_n] += t - last
    cur_n += dlt; last = t
tot = sum(hist.values())
print("span %.1f ms, at least one in flight %.1f ms" % (tot / 1e6, (tot - hist.get(0, 0)) / 1e6))
print("in flight (share of the span):", " ".join(f"{k}:{100*v/tot:.1f}%" for k, v in sorted(hist.items())))
# per stream: gap between the end of one dispatch and the start of the next
if c_q:
    i = sel.index(c_q)
    lastend = {}
    gaps = []
    for r in rows:
        if pat in r[0]:
            if r[i] in lastend:
                gaps.append((r[1] - lastend[r[i]]) / 1e3)
            lastend[r[i]] = r[2]
    gaps.sort()
    if gaps:
        print(f"gap to the previous dispatch of the same queue: p10={gaps[len(gaps)//10]:.1f} p50={gaps[len(gaps)//2]:.1f} p90={gaps[9*len(gaps)//10]:.1f} us")
