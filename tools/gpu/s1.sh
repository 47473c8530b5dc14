#!/bin/bash
# round-5 session 1: streaming LayerNorm tests + sweep, attention baseline
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; cd $root
timeout 900 python3 -m pytest tests/test_ops_gpu.py -k "layernorm" -x -q 2>&1 | tail -15 > $out/s1_lntest.txt
cat $out/s1_lntest.txt
timeout 1200 python3 tools/ln_bench.py --sweep > $out/r05_ln_sweep.txt 2>&1
tail -5 $out/r05_ln_sweep.txt
timeout 300 python3 tools/attn_bench.py --student-rows 768 --student-s 152 > $out/s1_attn.txt 2>&1
cat $out/s1_attn.txt
