#!/bin/bash
# round 6: the teacher's qkv product forms the row-norm maxima of its attention's band limit (ABI 14, A2V_EPI_QKNORM): tests, then the
# whole step with the switch off / on (alternating)
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
out=$root/gpurun_out/r06_qkn.txt; : > $out
export PYTHONUNBUFFERED=1
timeout 1200 python3 -m pytest tests/test_ops_gpu.py -x -q -k "qknorm or attention or linear or p8 or gemm" 2>&1 | tail -12 | tee -a $out
bash tools/gpu/ab_env.sh ${1:-4} "A2V_QKNORM=0" "-" > /dev/null 2>&1
cat gpurun_out/ab_env.txt | tee -a $out
