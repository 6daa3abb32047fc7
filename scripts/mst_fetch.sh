#!/bin/bash
# Dev tool (GPU box): what the Boruvka passes of one bench step fetch (FETCH_SIZE, its own rocprofv3 --pmc pass) and how
# long they take (kernel trace) -> gpurun_out/mst_fetch.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
{
$ROOT/scripts/mst_trace.sh | tail -14
rm -rf /tmp/prof_mf; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_mf -o run -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>&1
python $ROOT/scripts/rocpd_summary.py $(find /tmp/prof_mf -name "*.db") | grep -E "boruvka_(row|col)"
} > $OUT/mst_fetch.txt 2>&1
cat $OUT/mst_fetch.txt
