#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs: per-kernel time stats and PMC counter sums.
usage: rocpd_summary.py <results.db> [...]   -> prints a compact text summary"""
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 90 else name[:87] + "..."


for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"== {path}")
    try:
        rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        print("kernel-trace stats (durations in us):")
        print(f"{'calls':>6} {'total_us':>16} {'avg_us':>16} {'pct':>7}  name")
        for name, calls, tot, avg, pct in rows[:12]:
            print(f"{calls:>6} {tot:>16.0f} {avg:>16.1f} {pct:>7.2f}  {short(name)}")
    except sqlite3.Error as e:
        print("no top_kernels:", e)
    try:
        cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        q = ("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
             "group by kernel_name, counter_name order by kernel_name, counter_name")
        rows = list(cur.execute(q))
        if rows:
            print("PMC counters (per kernel: dispatches, sum, mean per dispatch):")
            for k, c, n, s, a in rows:
                if "lcsgpu" in k:
                    print(f"  {short(k):60s} {c:24s} n={n:<4d} sum={s:<22.0f} mean={a:.1f}")
    except sqlite3.Error as e:
        print("no counters:", e)
