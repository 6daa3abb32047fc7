#!/bin/bash
# round 5, GPU call L: what the driver runs at round end, on the final code: the GPU suite, smoke(), the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=10 ) > gpurun_out/l_suite.txt 2>&1
tail -22 gpurun_out/l_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py > gpurun_out/l_bench.json 2> gpurun_out/l_bench.err; cat gpurun_out/l_bench.json | cut -c1-400
