#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_x3_ab.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "x3 or band" 2>&1 | tail -3 | tee -a $out
for i in 1 2; do
  A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_prev.so timeout 300 python3 tools/x3_bench.py 2>&1 | grep "parity mode" | sed 's/^/prev /' | tee -a $out
  timeout 300 python3 tools/x3_bench.py 2>&1 | grep "parity mode" | sed 's/^/new  /' | tee -a $out
  A2V_DEBUG_SET=19=0 timeout 300 python3 tools/x3_bench.py 2>&1 | grep "parity mode" | sed 's/^/new, band limit off  /' | tee -a $out
done
