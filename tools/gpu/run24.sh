#!/bin/bash
# 8-wave sinc_fused_bwd: parity tests, then its rate (tools/elem_bench.py)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "sinc" > gpurun_out/t24_ops.log 2>&1
echo "ops rc=$?" > gpurun_out/t24_rc.log
timeout 600 python tools/elem_bench.py > gpurun_out/t24_elem.log 2>&1
tail -n 3 gpurun_out/t24_ops.log; grep -i sinc gpurun_out/t24_elem.log; cat gpurun_out/t24_rc.log
