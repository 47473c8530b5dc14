#!/bin/bash
# round 6: NT GEMM with two 32-MFMA phases per K-tile.  First form (source not kept; the "lib=" arm of the first log section): 64 x 64 wave
# tiles, one B half per wave -- slower, mostly through the epilogue's new column ownership.  Second form (-DP8_TWO_PHASE, variant p8tp): the
# four-phase form's column ownership and epilogue, both B halves' fragments read in phase 1.
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
out=$root/gpurun_out/r06_nt2b.txt; : > $out
export PYTHONUNBUFFERED=1
A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_p8tp.so timeout 1500 python3 -m pytest tests/test_ops_gpu.py -x -q -k "linear or p8 or gemm or dact or qknorm" 2>&1 | tail -3 | tee -a $out
for l in "" p8tp "" p8tp; do
  if [ -z "$l" ]; then unset A2V_LIB_PATH; else export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_$l.so; fi
  echo "## lib=$l" | tee -a $out; timeout 300 python3 tools/p8_probe.py 2>&1 | grep -v "amdgpu.ids\|per tile\|stamp" | tee -a $out
done
unset A2V_LIB_PATH
bash tools/gpu/ab_env.sh 3 "-" "A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_p8tp.so" > /dev/null 2>&1
cat gpurun_out/ab_env.txt | tee -a $out
