#!/bin/bash
# round 6: decoder convolution, in-order front (default) against the older per-clip chunks (a2v_debug_set(38, -1)) and other front widths
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
out=$root/gpurun_out/r06_dec.txt; : > $out
export PYTHONUNBUFFERED=1
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "decoder or conv_dec or dec" 2>&1 | tail -3 | tee -a $out
for s in "38=-1" "" "38=64" "38=256" "38=-1" ""; do
  if [ -z "$s" ]; then unset A2V_DEBUG_SET; else export A2V_DEBUG_SET=$s; fi
  echo "## A2V_DEBUG_SET=$s" | tee -a $out
  timeout 300 python3 tools/dec_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $out
done
unset A2V_DEBUG_SET
bash tools/gpu/ab_env.sh 3 "A2V_DEBUG_SET=38=-1" "-" > /dev/null 2>&1
cat gpurun_out/ab_env.txt | tee -a $out
