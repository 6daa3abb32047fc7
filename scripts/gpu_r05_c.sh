#!/bin/bash
# round 5, GPU call C: the CLI flags, pool / groups / stage sweeps of C5 at 3 000 000 sequences, the bench on the ragged
# workloads and with --pmc, UPGMA at 250 000 sequences, the whole suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_endtoend.py -x -q -m gpu -k "round_5 or undefined" > gpurun_out/c_quick.txt 2>&1; echo "rc=$?" >> gpurun_out/c_quick.txt
tail -4 gpurun_out/c_quick.txt
F=/tmp/family_3000000_300.fasta
python - <<PY
import sys, os
sys.path.insert(0, '.')
from famsa_amd import seqio
if not os.path.exists("$F"):
    seqio.family_fasta(3000000, 300, "$F")
PY
: > gpurun_out/c_c5_sweep.txt
run() { # label, env...
  label=$1; shift
  env "$@" famsa_amd/famsa-gpu -v -medoidtree -gt upgma -gt_export $F /tmp/sw.dnd 2> /tmp/sw.err
  echo "$label $(grep -E 'time.tree_build|gpu.lcs_kernel_ms' /tmp/sw.err | tr '\n' ' ') sha=$(sha256sum /tmp/sw.dnd | cut -c1-12)" >> gpurun_out/c_c5_sweep.txt
}
for rep in 1 2; do
  run "default(pool=32,groups=4,stage0=16,look=16)" X=1
  run "pool=48" FAMSA_HOST_TEST=pool=48
  run "pool=64" FAMSA_HOST_TEST=pool=64
  run "pool=64,groups=8" FAMSA_HOST_TEST=pool=64 LCSGPU_TUNE=clarans_groups=8
  run "pool=96,groups=6" FAMSA_HOST_TEST=pool=96 LCSGPU_TUNE=clarans_groups=6
  run "pool=64,groups=2" FAMSA_HOST_TEST=pool=64 LCSGPU_TUNE=clarans_groups=2
  run "stage0=32" LCSGPU_TUNE=clarans_stage0=32
  run "stage0=8" LCSGPU_TUNE=clarans_stage0=8
  run "look=32" LCSGPU_TUNE=clarans_look=32
  run "look=8" LCSGPU_TUNE=clarans_look=8
  run "share=0" LCSGPU_TUNE=lcs_share_lds=0
  run "pool=64,share=0" FAMSA_HOST_TEST=pool=64 LCSGPU_TUNE=lcs_share_lds=0
done
cat gpurun_out/c_c5_sweep.txt
: > gpurun_out/c_bench_workloads.txt
for w in "family sorted" "family input" "realmix sorted" "realmix input"; do
  set -- $w
  python bench.py --workload $1 --order $2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/c_bench_workloads.txt
done
python - <<'PY'
import json
for ln in open("gpurun_out/c_bench_workloads.txt"):
    d = json.loads(ln)
    print(d["config"]["workload"][:70], "|", round(d["value"]), d["unit"], "ms/step", round(d["ms_per_step"], 2), "kernel_ms", round(d["roofline"]["kernel_ms"], 2), "valu", round(d["roofline"]["valu"]["frac"], 3), "hbm", round(d["roofline"]["frac"], 3))
PY
python bench.py --pmc --steps 3 --warmup 1 > gpurun_out/c_bench_pmc.txt 2> gpurun_out/c_bench_pmc.err
tail -1 gpurun_out/c_bench_pmc.txt | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_source'], d['roofline']['frac'], d['roofline']['valu']['frac'], d.get('cpu_baseline', {}).get('value'))"
timeout 600 python scripts/upgma_beyond.py 250000 400 2>&1 | grep -v "^lcsgpu_create" | tail -12
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/c_suite.txt 2>&1
tail -8 gpurun_out/c_suite.txt
