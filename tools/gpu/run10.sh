set -x
mkdir -p gpurun_out
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; tail -c 2500 gpurun_out/r03_bench_default.json
python bench.py --gemm-table gpurun_out/gemm_table_r03.json --steps 2 --warmup 1 --cpu-baseline skip --parity-leg skip --global-batch 64 > gpurun_out/r03_bench_gb64.json 2>&1
bash tools/gpu/run_profiles.sh r03 2>&1 | grep -v "simple_timer" | tail -12
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp_x3 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_x3 -- python3 $GRAFT_REPO_ROOT/tools/x3_bench.py > /tmp/cp_x3.log 2>&1; tail -2 /tmp/cp_x3.log; f=$(find /tmp/cp_x3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r03_large_x3_kernel_stats.csv
