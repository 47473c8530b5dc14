#!/bin/bash
# full GPU suite + smoke + default bench (state check after the sinc_fused_bwd / bf16x3 conv changes)
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/t25_gpu.log 2>&1
echo "gpu rc=$?" > gpurun_out/t25_rc.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/t25_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/t25_rc.log
timeout 1200 python bench.py > gpurun_out/t25_bench.json 2> gpurun_out/t25_bench.err
echo "bench rc=$?" >> gpurun_out/t25_rc.log
tail -n 3 gpurun_out/t25_gpu.log; tail -n 1 gpurun_out/t25_smoke.log; cat gpurun_out/t25_rc.log; tail -c 600 gpurun_out/t25_bench.json
