#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_p8_ab4.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
timeout 600 python3 -m pytest tests/test_ops_gpu.py -x -q -k "p8 or linear or gemm or dact" 2>&1 | tail -3 | tee -a $out
for lib in prev "" prev ""; do
  if [ -z "$lib" ]; then unset A2V_LIB_PATH; else export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_$lib.so; fi
  timeout 120 python3 tools/p8_probe.py --only fc1 2>&1 | grep -v amdgpu.ids | tee -a $out
done
unset A2V_LIB_PATH
bash tools/gpu/ab_bench.sh 3 > /dev/null 2>&1; cat gpurun_out/ab_bench.txt | tee -a $out
