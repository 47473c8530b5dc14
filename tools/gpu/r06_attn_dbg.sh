#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
out=gpurun_out/r06_attn_dbg.txt; : > $out
for f in ${FLAGS:-0 256 512 768 1024 1536}; do
  echo "flags=$f $(python3 tools/attn_bench.py --which student --student-rows 768 --student-s 152 --iters 20 --debug-set 40=$f | tail -1)" | tee -a $out
done
