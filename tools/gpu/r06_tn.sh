#!/bin/bash
# round 6: what bounds the K loop of the weight-gradient GEMM (timing builds -DT8_EXP=n) + the upper bound of a cheaper GELU
# (-DA2V_EXP_NOGELU) in the step.   raw log -> gpurun_out/r06_tn.txt
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
out=$root/gpurun_out/r06_tn.txt; : > $out
export PYTHONUNBUFFERED=1
lib() { if [ -z "$1" ]; then unset A2V_LIB_PATH; else export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_$1.so; fi; }
for l in "" t8e1 t8e2 t8e3 t8e4 ""; do
  [ -n "$l" ] && [ ! -f $root/animal2vec_amd/liba2v_hip_$l.so ] && continue
  lib "$l"; echo "## lib=$l" | tee -a $out; timeout 300 python3 tools/tn_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $out
done
lib ""
if [ -f $root/animal2vec_amd/liba2v_hip_nogelu.so ]; then
  bash tools/gpu/ab_env.sh 2 "-" "A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_nogelu.so" > /dev/null 2>&1
  cat gpurun_out/ab_env.txt | tee -a $out
  A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_nogelu.so bash tools/collect_profiles.sh r06nogelu large "--steps 2 --warmup 1 --global-batch 64" "" | tail -1
fi
