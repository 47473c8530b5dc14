#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; cd $root
timeout 600 python3 -m pytest tests/test_ops_gpu.py -k "conv" -x -q 2>&1 | tail -3
timeout 600 python3 tools/dec_bench.py > $out/r05_dec_bench.txt 2>&1; cat $out/r05_dec_bench.txt
: > $out/s7_ab.txt
for i in 1 2; do
  for o in 4 0; do
    v=$(A2V_DEBUG_SET="35=$o" python3 bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip 2>/dev/null | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
    echo "order=$o $v" | tee -a $out/s7_ab.txt
  done
done
