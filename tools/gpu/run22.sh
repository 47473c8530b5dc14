set -x
mkdir -p gpurun_out
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; tail -c 1200 gpurun_out/r03_bench_default.json
bash tools/gpu/run_profiles.sh r03 2>&1 | grep -v "simple_timer" | tail -6
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp_x3 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_x3 -- python3 $GRAFT_REPO_ROOT/tools/x3_bench.py > /tmp/cp_x3.log 2>&1; tail -1 /tmp/cp_x3.log; f=$(find /tmp/cp_x3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r03_large_x3_kernel_stats.csv
cd $GRAFT_REPO_ROOT && bash tools/collect_profiles.sh r03 large_mb16 "--steps 6 --warmup 2 --global-batch 16" "" 2>&1 | tail -1
python bench.py --model base --global-batch 64 --steps 3 --warmup 1 --cpu-baseline skip --parity-leg skip > gpurun_out/r03_bench_base.json 2>&1; tail -c 200 gpurun_out/r03_bench_base.json
python bench.py --model large44k --samples 441000 --global-batch 4 --micro-batch 2 --steps 2 --warmup 1 --cpu-baseline skip --parity-leg skip > gpurun_out/r03_bench_large44k.json 2>&1; tail -c 200 gpurun_out/r03_bench_large44k.json
python tools/finetune_bench.py --steps 2 --warmup 1 --accum 2 2>&1 | tail -3
