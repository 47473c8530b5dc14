#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_w2_exp.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
export A2V_DEBUG_SET=25=2
for lib in "" w2e1 w2e2 w2e4 w2e7; do
  if [ -z "$lib" ]; then unset A2V_LIB_PATH; else export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_$lib.so; fi
  echo "## lib=$lib" | tee -a $out
  for sh in qkv "fc2   bias"; do timeout 120 python3 tools/p8_probe.py --only "$sh" 2>&1 | grep -v "amdgpu.ids\|^#" | tee -a $out; done
done
