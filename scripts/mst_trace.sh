#!/bin/bash
# Dev tool (GPU box): per-launch durations of the Boruvka passes of one bench step, in launch order.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_mst; rocprofv3 --kernel-trace -d /tmp/prof_mst -o run -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity "$@" > /dev/null 2>&1
python - $(find /tmp/prof_mst -name "*.db") <<'EOP'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
t = kd[0]
cols = [d[1] for d in cur.execute(f"pragma table_info({t})")]
name = "name" if "name" in cols else "kernel_name"
rows = list(cur.execute(f"select {name}, start, end from {t} order by start"))
r = c = 0
for n, s, e in rows:
    if "boruvka_row" in n or "boruvka_col" in n:
        print(("row" if "boruvka_row" in n else "col"), f"{(e - s) / 1e3:9.1f} us")
tot = {}
for n, s, e in rows:
    k = n.split("(")[0].replace("void ", "")[:60]
    tot[k] = tot.get(k, 0) + (e - s) / 1e3
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:12]:
    print(f"{v:12.1f} us  {k}")
EOP
