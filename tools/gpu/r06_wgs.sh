#!/bin/bash
# round 6: pos-conv weight gradient with its kk tiles split over two workgroups per CU (a2v_debug_set(40, v)): tests, the launch alone, the step
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
out=$root/gpurun_out/r06_wgs.txt; : > $out
export PYTHONUNBUFFERED=1
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "wgrad or conv" 2>&1 | tail -3 | tee -a $out
for s in 40=0 40=1 40=0 40=1; do
  echo "## debug-set $s" | tee -a $out
  timeout 300 python3 tools/gemm_bench.py --only "posconv wgrad" --debug-set $s --reps 10 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $out
done
bash tools/gpu/ab_env.sh 3 "A2V_DEBUG_SET=40=0" "-" > /dev/null 2>&1
cat gpurun_out/ab_env.txt | tee -a $out
