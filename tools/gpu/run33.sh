#!/bin/bash
# bf16x3: images kept from the forward for the backward's weight gradients: parity tests, then the mode's rate
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "x3" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_pretrain_gpu.py -m gpu -x -q -k "bf16x3 or x3 or large_full_size" 2>&1 | tail -2
timeout 600 python tools/x3_bench.py 2>&1 | tail -1
timeout 600 python tools/x3_bench.py 2>&1 | tail -1
