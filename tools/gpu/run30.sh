#!/bin/bash
# end-of-round evidence (second half of round 3): default bench line, kernel stats + PMC passes of the headline config, the 8- / 16-clip
# steps, the bf16x3 mode's kernel table, element-wise bench
set -x
mkdir -p gpurun_out
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; tail -c 600 gpurun_out/r03_bench_default.json
bash tools/collect_profiles.sh r03e large "--steps 2 --warmup 1 --global-batch 64" "--steps 1 --warmup 0 --global-batch 64" 2>&1 | grep -v simple_timer | tail -3
bash tools/collect_profiles.sh r03e large_mb8 "--steps 6 --warmup 2 --global-batch 8" "" 2>&1 | tail -1
bash tools/collect_profiles.sh r03e large_mb16 "--steps 6 --warmup 2 --global-batch 16" "" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp_x3 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_x3 -- python3 $GRAFT_REPO_ROOT/tools/x3_bench.py > /tmp/cp_x3.log 2>&1; tail -1 /tmp/cp_x3.log; f=$(find /tmp/cp_x3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r03e_large_x3_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python tools/elem_bench.py > gpurun_out/r03e_elem_bench.log 2>&1; grep -i "sinc\|instnorm" gpurun_out/r03e_elem_bench.log
for gb in 8 16; do python bench.py --global-batch $gb --steps 6 --warmup 2 --cpu-baseline skip --parity-leg skip --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('gb$gb', d['value'], d['ms_per_step'])"; done
