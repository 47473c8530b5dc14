#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; cd $root
timeout 1200 python3 -m pytest tests/test_ops_gpu.py -k "dact8 or p8 or linear_plain or multi_round" -x -q 2>&1 | tail -6 > $out/s4_test.txt
cat $out/s4_test.txt
: > $out/s4_ab.txt
for i in 1 2; do
  for d8 in 0 1; do
    v=$(A2V_DACT8=$d8 python3 bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip 2>/dev/null | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
    echo "dact8=$d8 $v" | tee -a $out/s4_ab.txt
  done
done
