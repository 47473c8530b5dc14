mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp_x3 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_x3 -- python3 $GRAFT_REPO_ROOT/tools/x3_bench.py > /tmp/cp_x3.log 2>&1; tail -1 /tmp/cp_x3.log; f=$(find /tmp/cp_x3 -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-150
