#!/bin/bash
# GPU box: package power and shader clock sampled every ~0.5 s while the default bench workload runs
# -> gpurun_out/power_clock_<tag>.txt (the power-limit claim of DESIGN section 4 rests on this).  usage: scripts/power_sample.sh r04
TAG=${1:-rXX}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/power_clock_${TAG}.txt
{
echo "# rocm-smi --showpower --showclocks sampled every ~0.5 s while \`python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-parity\` ran (round ${TAG})"
echo "# sclk_MHz  package_W"
} > $OUT
( python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/bench_power_${TAG}.json 2>/dev/null ) &
BPID=$!
while kill -0 $BPID 2>/dev/null; do
  S=$(rocm-smi --showpower --showclocks 2>/dev/null)
  W=$(echo "$S" | grep -iE "Package Power|Socket Power" | head -1 | grep -oE "[0-9]+\.[0-9]+" | head -1)
  C=$(echo "$S" | grep -iE "sclk clock level" | head -1 | grep -oE "\(([0-9]+)Mhz\)" | grep -oE "[0-9]+")
  echo "  ${C:-?}    ${W:-?}" >> $OUT
  sleep 0.5
done
wait $BPID
echo "# bench line of this run:" >> $OUT
sed 's/^/# /' gpurun_out/bench_power_${TAG}.json | cut -c1-400 >> $OUT
cat $OUT | tail -40
