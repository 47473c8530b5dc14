#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_x3_ab2.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "x3" 2>&1 | tail -3 | tee -a $out
timeout 900 python3 -m pytest tests/test_pretrain_gpu.py -x -q -k "x3" 2>&1 | tail -3 | tee -a $out
for i in 1 2; do
  A2V_X3_OUT_IMAGE=0 timeout 300 python3 tools/x3_bench.py 2>&1 | grep "parity mode" | sed 's/^/images off /' | tee -a $out
  timeout 300 python3 tools/x3_bench.py 2>&1 | grep "parity mode" | sed 's/^/new        /' | tee -a $out
done
