#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; cd $root
timeout 900 python3 -m pytest tests/test_ops_gpu.py -k "layernorm" -x -q 2>&1 | tail -5 > $out/s2_lntest.txt
cat $out/s2_lntest.txt
timeout 1200 python3 tools/ln_bench.py --front > $out/r05_ln_front.txt 2>&1
tail -3 $out/r05_ln_front.txt
