#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_x3_ab3.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
for i in 1 2 3; do
  A2V_X3_DH_IMAGE=0 timeout 300 python3 tools/x3_bench.py 2>&1 | grep "parity mode" | sed 's/^/dh image off /' | tee -a $out
  timeout 300 python3 tools/x3_bench.py 2>&1 | grep "parity mode" | sed 's/^/dh image on  /' | tee -a $out
done
