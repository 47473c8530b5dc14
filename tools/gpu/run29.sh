#!/bin/bash
# streaming stores of the persistent NT GEMM (debug key 23: outputs >= v MB; 0 = never): microbench + whole steps, same box
for r in 0 400 200; do
  echo "== nt_mb $r"
  A2V_DEBUG_SET="23=$r" timeout 120 python tools/nt_gemm_check.py 2>&1 | grep "N  3072\|N  1024 K  1024\|K  4096\|K   512"
done
B="python bench.py --steps 3 --warmup 1 --parity-leg skip --cpu-baseline skip --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], d["value"], d["ms_per_step"])'
for i in 1 2; do
  for r in 0 400 200 900 100000; do A2V_DEBUG_SET="23=$r" $B 2>/dev/null | python -c "$P" nt_mb$r; done
done
