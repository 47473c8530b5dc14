#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_p8_ab2.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
for lib in prev "" pref prev "" pref; do
  if [ -z "$lib" ]; then unset A2V_LIB_PATH; else export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_$lib.so; fi
  timeout 300 python3 tools/p8_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $out
done
A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_prefprobe.so timeout 300 python3 tools/p8_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $out
