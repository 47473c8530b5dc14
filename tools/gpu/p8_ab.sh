#!/bin/bash
# same-box A/B of the persistent NT GEMM: previous build (tools/build_prev.sh <rev>) vs the current one, then the per-tile stamps
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_p8_ab.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
timeout 600 python3 -m pytest tests/test_ops_gpu.py -x -q -k "p8 or linear or gemm" 2>&1 | tail -3 | tee -a $out
A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_prev.so timeout 300 python3 tools/p8_probe.py 2>&1 | tee -a $out
unset A2V_LIB_PATH
timeout 300 python3 tools/p8_probe.py 2>&1 | tee -a $out
A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_probe.so timeout 300 python3 tools/p8_probe.py 2>&1 | tee -a $out
