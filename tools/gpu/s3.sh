#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; cd $root
timeout 900 python3 -m pytest tests/test_ops_gpu.py -k "layernorm" -x -q 2>&1 | tail -3 > $out/s3_lntest.txt
cat $out/s3_lntest.txt
timeout 600 python3 tools/ln_bench.py > $out/r05_ln_policy.txt 2>&1
cat $out/r05_ln_policy.txt
bash tools/gpu/ab_bench.sh 2
