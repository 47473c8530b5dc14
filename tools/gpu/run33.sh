#!/bin/bash
# one-pass bf16x3 slab conv: ops tests, abi tests, the full-size bf16x3 parity test, then the mode's rate
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_abi.py -m gpu -x -q -k "x3 or abi or symbols" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_pretrain_gpu.py -m gpu -x -q -k "bf16x3 or x3" 2>&1 | tail -2
timeout 600 python tools/x3_bench.py 2>&1 | tail -1
A2V_X3_CONV_FUSED=0 timeout 600 python tools/x3_bench.py 2>&1 | tail -1
timeout 600 python tools/x3_bench.py 2>&1 | tail -1
