#!/bin/bash
# Round-4 probe of the persistent NT GEMM's tile seam (tools/p8_probe.py): product, L2-resident stores, start skew, per-tile stamps.
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_p8_probe.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
unset A2V_LIB_PATH
timeout 300 python3 tools/p8_probe.py 2>&1 | tee -a $out
A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_l2st.so timeout 300 python3 tools/p8_probe.py 2>&1 | tee -a $out
A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_skew.so timeout 400 python3 tools/p8_probe.py --skews 0,1,2,4,6 2>&1 | tee -a $out
A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_probe.so timeout 300 python3 tools/p8_probe.py 2>&1 | tee -a $out
