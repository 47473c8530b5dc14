#!/bin/bash
# branch-free staging in the slab conv / wgrad-slab kernels: conv parity tests, pretrain parity, then the default bench
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv or slab or wgrad" > gpurun_out/t26_ops.log 2>&1
echo "ops rc=$?" > gpurun_out/t26_rc.log
timeout 1500 python -m pytest tests/test_pretrain_gpu.py -m gpu -x -q > gpurun_out/t26_pre.log 2>&1
echo "pre rc=$?" >> gpurun_out/t26_rc.log
timeout 900 python bench.py --parity-leg skip --cpu-baseline skip > gpurun_out/t26_bench.json 2> gpurun_out/t26_bench.err
echo "bench rc=$?" >> gpurun_out/t26_rc.log
tail -n 2 gpurun_out/t26_ops.log; tail -n 2 gpurun_out/t26_pre.log; cat gpurun_out/t26_rc.log; head -c 300 gpurun_out/t26_bench.json
