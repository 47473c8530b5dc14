#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_w2_ab.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
A2V_DEBUG_SET=25=2 timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "p8 or linear or gemm or dact" 2>&1 | tail -5 | tee -a $out
for s in "" 25=2 "" 25=2; do
  if [ -z "$s" ]; then unset A2V_DEBUG_SET; else export A2V_DEBUG_SET=$s; fi
  echo "## A2V_DEBUG_SET=$s" | tee -a $out
  timeout 300 python3 tools/p8_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $out
done
