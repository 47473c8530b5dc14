#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_dma_aux.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
for lib in "" aux1 aux2 aux4 aux6; do
  if [ -z "$lib" ]; then unset A2V_LIB_PATH; else export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_$lib.so; fi
  for s in "" 25=2; do
    if [ -z "$s" ]; then unset A2V_DEBUG_SET; else export A2V_DEBUG_SET=$s; fi
    echo "## lib=$lib set=$s" | tee -a $out
    for sh in qkv "fc2   bias"; do timeout 120 python3 tools/p8_probe.py --only "$sh" 2>&1 | grep -v "amdgpu.ids\|^#" | tee -a $out; done
  done
done
