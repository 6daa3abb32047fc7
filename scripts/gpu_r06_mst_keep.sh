#!/bin/bash
# round 6 (second session): the MST passes -- the multiplication before the division (mst_crossmul), the records that stand
# from the round before (mst_keep), the survivors' fetches in groups (mst_grouped): tests first, then the 13 774-record set,
# the ragged family set and 100 000 x 400 aa with each switch off in turn
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out
OUT=gpurun_out/mst_keep_r06.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_realmix.py tests/test_gpu_sharded_mst.py tests/test_gpu_fused_mst.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_atsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee -a $OUT
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 $2 step', round(d['ms_per_step'],3), 'ms; LCS', round(d['roofline']['kernel_ms'],3), 'ms; MST', round(d['mst']['ms_per_step'],3), d['mst']['rounds'], d['mst']['edges_sha256'][:12])"
}
for tune in "mst_keep=0,mst_crossmul=0,mst_grouped=0" "mst_grouped=0" "mst_keep=0" "mst_crossmul=0" "all=on"; do
  for wl in realmix family; do
    LCSGPU_TUNE=$tune python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --pmc off 2>/dev/null | line $tune $wl | tee -a $OUT
  done
  LCSGPU_TUNE=$tune python bench.py --steps 5 --warmup 2 --no-cpu-baseline --pmc off 2>/dev/null | line $tune 100000x400 | tee -a $OUT
done
cd /tmp && export TMPDIR=/tmp
for wl in realmix family; do
rm -rf /tmp/prof_small; rocprofv3 --kernel-trace --stats -d /tmp/prof_small -o run -- python $ROOT/bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline --no-parity --pmc off > /dev/null 2>&1
echo "== $wl, defaults" | tee -a $ROOT/$OUT
python $ROOT/scripts/rocpd_summary.py $(find /tmp/prof_small -name "*.db") | head -16 | tee -a $ROOT/$OUT
done
