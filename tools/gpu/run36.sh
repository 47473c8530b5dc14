#!/bin/bash
# bf16x3: fc1's hidden activation as its split image only (A2V_EPI_X3IMG): op tests, pretrain x3 parity, the mode's rate with / without
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "x3" 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_pretrain_gpu.py -m gpu -x -q -k "bf16x3 or x3 or large_full_size" 2>&1 | tail -2
timeout 600 python tools/x3_bench.py 2>&1 | tail -1
A2V_X3_OUT_IMAGE=0 timeout 600 python tools/x3_bench.py 2>&1 | tail -1
timeout 600 python tools/x3_bench.py 2>&1 | tail -1
