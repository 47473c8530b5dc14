#!/bin/bash
# final state: full GPU suite, smoke, default bench line, bf16x3 kernel table
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/t34_gpu.log 2>&1
echo "gpu rc=$?" > gpurun_out/t34_rc.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/t34_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/t34_rc.log
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
echo "bench rc=$?" >> gpurun_out/t34_rc.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp_x3 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_x3 -- python3 $GRAFT_REPO_ROOT/tools/x3_bench.py > /tmp/cp_x3.log 2>&1; tail -1 /tmp/cp_x3.log; f=$(find /tmp/cp_x3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r03f_large_x3_kernel_stats.csv
cd $GRAFT_REPO_ROOT
tail -n 3 gpurun_out/t34_gpu.log; tail -n 1 gpurun_out/t34_smoke.log; cat gpurun_out/t34_rc.log; tail -c 500 gpurun_out/r03_bench_default.json
