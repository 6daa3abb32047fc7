#!/bin/bash
# round 6 (second session): the partner's first chunk kept across the ref groups of a workgroup for refs of <= 6 half-words
# (LCS_FIRST_KEPT_H; variant "nokeep" = built with -DLCS_FIRST_KEPT_H=0): parity first, then the step on short sets
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
R=gpurun_out/first_chunk_r06.txt
: > $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_realmix.py tests/test_gpu_fused_mst.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee -a $R
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 $2 step', round(d['ms_per_step'],3), 'ms; LCS', round(d['roofline']['kernel_ms'],3), 'ms; valu', round(d['roofline']['valu']['frac'],3), '; Tcell/s', round(d['value']/1e3,1))"
}
for rep in 1 2; do
for v in keep nokeep; do
  LIB=famsa_amd/liblcsgpu.so; [ $v = nokeep ] && LIB=famsa_amd/_variants/nokeep/liblcsgpu.so
  LCSGPU_LIB=$LIB python bench.py --workload realmix --steps 20 --warmup 3 --no-cpu-baseline --pmc off 2>/dev/null | line $v realmix | tee -a $R
  for len in 40 70 100 150 190; do
    LCSGPU_LIB=$LIB python bench.py --n-seqs 40000 --seq-len $len --steps 5 --warmup 2 --no-cpu-baseline --pmc off 2>/dev/null | line $v 40000x$len | tee -a $R
  done
done
done
for v in keep nokeep; do
  LIB=famsa_amd/liblcsgpu.so; [ $v = nokeep ] && LIB=famsa_amd/_variants/nokeep/liblcsgpu.so
  LCSGPU_LIB=$LIB python bench.py --steps 3 --warmup 1 --no-cpu-baseline --pmc off 2>/dev/null | line $v 100000x400 | tee -a $R
done
