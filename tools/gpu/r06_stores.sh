#!/bin/bash
# round 6: cache-policy bits on the NT GEMM's epilogue stores (experiment builds -DP8_ST_AUX=n: st0 plain, st1 nt, st5 nt sc0, st6 nt sc0 sc1;
# the product is 4 = nt sc1; stall = -DA2V_STREAM_ST, every 16-byte bf16 store of every kernel as nt sc1) on the step's five shapes,
# alternating on one box, then the whole step.   raw log -> gpurun_out/r06_stores.txt
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
out=$root/gpurun_out/r06_stores.txt; : > $out
export PYTHONUNBUFFERED=1
lib() { if [ -z "$1" ]; then unset A2V_LIB_PATH; else export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_$1.so; fi; }
probe() { timeout 300 python3 tools/p8_probe.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a $out; }
lib ""
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "p8 or linear or gemm or dact" 2>&1 | tail -3 | tee -a $out
for l in st0 "" st1 st5 st6 st0 ""; do
  [ -n "$l" ] && [ ! -f $root/animal2vec_amd/liba2v_hip_$l.so ] && continue
  lib "$l"; echo "## lib=$l" | tee -a $out; probe
done
lib ""
bash tools/gpu/ab_env.sh 3 "A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_st0.so" "-" "A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_stall.so" > /dev/null 2>&1
cat gpurun_out/ab_env.txt | tee -a $out
