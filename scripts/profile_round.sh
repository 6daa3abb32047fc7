#!/bin/bash
# The profiles of a round, on the GPU box: rocprofv3 kernel-trace statistics of the default bench.py command and
# the PMC passes (each in its own run, with --kernel-trace only -- MI355X_MICROARCH.md, HBM/rocprofv3 section).
# usage: scripts/profile_round.sh r02   -> gpurun_out/rocprof_<tag>_summary.txt, gpurun_out/pmc_<tag>.json
set -u
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
S=$OUT/rocprof_${TAG}_summary.txt
{
echo "# rocprofv3 summaries, round ${TAG}, MI355X, default bench workload (100 000 x 400 aa; step = LCS triangle + sharded-Boruvka MST)"
echo "# kernel-trace: rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity"
echo "# PMC passes:  rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity"
echo "# durations in MICROSECONDS (rocpd top_kernels); FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE reads 1/2 of a wide streaming read on gfx950"
} > "$S"
rm -rf /tmp/prof_kt; rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o run -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --pmc off > $OUT/bench_${TAG}_under_rocprof.json 2>/dev/null
python $ROOT/scripts/rocpd_summary.py $(find /tmp/prof_kt -name "*.db") >> "$S"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/prof_pmc$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/prof_pmc$i -o run -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity --pmc off > /dev/null 2>&1
  python $ROOT/scripts/rocpd_summary.py $(find /tmp/prof_pmc$i -name "*.db") >> "$S"
done
python - "$S" "$OUT/pmc_${TAG}.json" <<'EOP'
import json, re, sys
txt = open(sys.argv[1]).read()
out = {"_comment": "per launch of the hot kernel, default bench workload (n=100000 x 400 aa, 1 GPU), from separate rocprofv3 --pmc passes "
                   "(rocprof summary next to this file). traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024: FETCH_SIZE doubled per the "
                   "gfx950 correction of MI355X_MICROARCH.md", "n_seqs": 100000, "seq_len": 400, "n_gpus": 1}
for line in txt.splitlines():
    m = re.match(r"\s+lcsgpu::lcs_rows_kernel_pipe<13, 4, 4[^>]*>\S*\s+(\w+)\s+n=(\d+)\s+sum=(\d+)\s+mean=([\d.]+)", line)
    if m:
        out[m.group(1)] = float(m.group(4))
    m = re.match(r"\s+lcsgpu::(boruvka_\w+_kernel)<[^>]*>\S*\s+(FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE|SQ_INSTS_VALU|SQ_INSTS_SALU)\s+n=(\d+)\s+sum=(\d+)\s+mean=([\d.]+)", line)
    if m:
        out.setdefault(m.group(1), {})[m.group(2)] = float(m.group(5))
# which kernel and which library build these counters are about: bench.py passes the traffic figure on only to a run of the same
out["kernel"] = "lcsgpu::lcs_rows_kernel_pipe<13, 4, 4, false>"
try:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[2]))))
    import famsa_amd
    out["library"] = famsa_amd.load_library().lcsgpu_version().decode()
except Exception as e:
    out["library"] = None
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["traffic_bytes"] = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
EOP
