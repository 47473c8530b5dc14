#!/bin/bash
# bf16x3 parity mode, same-box A/B of its switches / of two builds on tools/x3_bench.py (8-clip optimizer steps):
#   bash tools/gpu/x3_ab.sh [rounds] "<env assignments of arm A>" "<env assignments of arm B>" ...   ("-" = none)  -> gpurun_out/x3_ab.txt
#   e.g.  bash tools/gpu/x3_ab.sh 3 A2V_X3_DH_IMAGE=0 -        bash tools/gpu/x3_ab.sh 2 A2V_LIB_PATH=$PWD/animal2vec_amd/liba2v_hip_prev.so - A2V_DEBUG_SET=19=0
rounds=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/x3_ab.txt
cd $root
: > $out
for i in $(seq 1 $rounds); do
  for arm in "$@"; do
    if [ "$arm" = "-" ]; then v=$(timeout 300 python3 tools/x3_bench.py 2>&1 | grep "parity mode")
    else v=$(env $arm timeout 300 python3 tools/x3_bench.py 2>&1 | grep "parity mode"); fi
    echo "[$arm] $v" | tee -a $out
  done
done
