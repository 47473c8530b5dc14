#!/bin/bash
# round 6: weight-gradient GEMM with two 32-MFMA phases per K-tile (64 x 64 wave tiles) against the four-phase form (variant tn4)
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
out=$root/gpurun_out/r06_tn2.txt; : > $out
export PYTHONUNBUFFERED=1
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "wgrad" 2>&1 | tail -3 | tee -a $out
for l in tn4 "" tn4 ""; do
  if [ -z "$l" ]; then unset A2V_LIB_PATH; else export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_$l.so; fi
  echo "## lib=$l" | tee -a $out; timeout 300 python3 tools/tn_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $out
done
unset A2V_LIB_PATH
bash tools/gpu/ab_env.sh 3 "A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_tn4.so" "-" > /dev/null 2>&1
cat gpurun_out/ab_env.txt | tee -a $out
