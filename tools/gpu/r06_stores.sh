#!/bin/bash
# round 6: cache-policy bits on the NT GEMM's epilogue stores (experiment builds -DP8_ST_AUX=n: stnt = nt, sta2 = sc1, sta3 = sc0 sc1,
# sta4 = nt sc1) against the product, on the step's five shapes; alternating, one box.   raw log -> gpurun_out/r06_stores.txt
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
out=$root/gpurun_out/r06_stores.txt; : > $out
export PYTHONUNBUFFERED=1
lib() { if [ -z "$1" ]; then unset A2V_LIB_PATH; else export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_$1.so; fi; }
probe() { timeout 300 python3 tools/p8_probe.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a $out; }
for l in "" stnt sta2 sta3 sta4 "" stnt sta2 sta3 sta4; do
  [ -n "$l" ] && [ ! -f $root/animal2vec_amd/liba2v_hip_$l.so ] && continue
  lib "$l"; echo "## lib=$l" | tee -a $out; probe
done
