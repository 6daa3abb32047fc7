cd /tmp && export TMPDIR=/tmp
for l in 1 0; do
  rm -rf /tmp/cp$l
  LCSGPU_CLARANS_LISTS=$l timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/cp$l -o run -- python $GRAFT_REPO_ROOT/scripts/clarans_bench.py 3 > /tmp/cp$l.log 2>&1
  echo "lists=$l"; tail -1 /tmp/cp$l.log
  python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $(find /tmp/cp$l -name "*.db") | head -8
done
