#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_p8_msweep.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_probe.so
for M in 8192 16384 32768 116736; do
  echo "## M=$M" | tee -a $out
  for sh in qkv "fc2   bias"; do timeout 120 python3 tools/p8_probe.py --M $M --only "$sh" 2>&1 | grep -v "amdgpu.ids\|^#\|epilogue starts" | tee -a $out; done
done
