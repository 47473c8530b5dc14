#!/bin/bash
# round 5: the 32 x 32 x 16 long-sequence attention forward against the 16 x 16 x 32 one (tests + micro-benchmark)
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
out=gpurun_out/s9_attn32.txt; : > $out
[ "$1" = "notest" ] || timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -15 | tee -a $out
for i in 1 2 3; do
  for s in "36=0" "36=1" "36=0 --debug-set 18=0" "36=1 --debug-set 18=0"; do
    echo "[$s] $(timeout 300 python tools/attn_bench.py --which teacher --iters 40 --debug-set $s 2>/dev/null | tail -1)" | tee -a $out
  done
done
