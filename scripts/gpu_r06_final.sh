#!/bin/bash
# round 6, the round's last code: the whole GPU suite, the contract line (PMC traffic by the run itself), the rocprofv3
# summaries + counters of the same command, the small set's step, the C5 runs, the end-to-end table
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/full_gpu_suite.txt
cat gpurun_out/full_gpu_suite.txt
python bench.py > gpurun_out/bench_r06_final.json 2> gpurun_out/bench_r06_final.err
tail -c 600 gpurun_out/bench_r06_final.json
bash scripts/profile_round.sh r06 > /dev/null 2>&1
bash scripts/gpu_r06_small.sh > gpurun_out/small_r06.txt 2>&1
bash scripts/gpu_r06_c5_runs.sh > /dev/null 2>&1
E2E_REFERENCE_FROM=profiles/e2e_r06.json timeout 1500 python scripts/e2e_compare.py r06 > gpurun_out/e2e_log.txt 2>&1
tail -3 gpurun_out/e2e_log.txt | cut -c1-400
