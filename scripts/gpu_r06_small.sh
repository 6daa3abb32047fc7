#!/bin/bash
# round 6: small problems -- the real 13 774-record set through bench.py (kernel trace) and hemopexin -gt nj
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out
python bench.py --workload realmix --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('realmix step', d['ms_per_step'], 'ms; LCS', d['roofline']['kernel_ms'], 'ms; MST', d['mst'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_small; rocprofv3 --kernel-trace --stats -d /tmp/prof_small -o run -- python $ROOT/bench.py --workload realmix --steps 5 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1
python $ROOT/scripts/rocpd_summary.py $(find /tmp/prof_small -name "*.db") | head -40
cd $ROOT
for i in 1 2 3; do famsa_amd/famsa-gpu -v -gt nj -gt_export tests/golden/hemopexin/hemopexin /tmp/nj.dnd 2>&1 | grep -E "tree_build"; done
python - <<PY
import hashlib, json
print("nj sha", hashlib.sha256(open("/tmp/nj.dnd","rb").read()).hexdigest()[:16])
PY
