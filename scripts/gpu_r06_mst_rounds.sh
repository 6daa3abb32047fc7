#!/bin/bash
# round 6 (second session): the whole GPU suite with the passes' kept records / multiplication test in, then the Boruvka
# rounds of the 13 774-record set launch by launch (one step), with and without the kept records
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/full_gpu_suite.txt
cat gpurun_out/full_gpu_suite.txt
OUT=$ROOT/gpurun_out/mst_rounds_r06.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for tune in "mst_keep=0,mst_crossmul=0" "mst_keep=1,mst_crossmul=1"; do
rm -rf /tmp/prof_small; LCSGPU_TUNE=$tune rocprofv3 --kernel-trace -d /tmp/prof_small -o run -- python $ROOT/bench.py --workload realmix --steps 1 --warmup 1 --no-cpu-baseline --no-parity --pmc off > /dev/null 2>&1
echo "== realmix, $tune: the launches of the last step" >> $OUT
python $ROOT/scripts/rocpd_launches.py $(find /tmp/prof_small -name "*.db") "" 0 | tail -110 | cut -c1-120 >> $OUT
done
