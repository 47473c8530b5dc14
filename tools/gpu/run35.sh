#!/bin/bash
# bf16x3 attention backward: op tests (timeout-guarded), pretrain x3 parity, the mode's rate with / without it
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "bwd_bf16x3" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "x3 or attention" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_pretrain_gpu.py -m gpu -x -q -k "bf16x3 or x3 or large_full_size" 2>&1 | tail -2
timeout 600 python tools/x3_bench.py 2>&1 | tail -1
A2V_X3_ATTN_BWD=0 timeout 600 python tools/x3_bench.py 2>&1 | tail -1
timeout 600 python tools/x3_bench.py 2>&1 | tail -1
