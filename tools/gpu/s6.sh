#!/bin/bash
# A/B of the grouped-conv launch orders (a2v_debug_set(35)): whole step, alternating
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; cd $root
timeout 600 python3 -m pytest tests/test_ops_gpu.py -k "conv" -x -q 2>&1 | tail -3
A2V_DEBUG_SET="35=3" timeout 600 python3 -m pytest tests/test_ops_gpu.py -k "conv" -x -q 2>&1 | tail -3
: > $out/s6_ab.txt
for i in 1 2; do
  for o in 0 1 2 3; do
    v=$(A2V_DEBUG_SET="35=$o" python3 bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip 2>/dev/null | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
    echo "order=$o $v" | tee -a $out/s6_ab.txt
  done
done
