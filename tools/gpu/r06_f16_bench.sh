#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
: > gpurun_out/r06_f16_ab.txt
for r in 1 2 3; do
  for f in 0 1; do
    v=$(A2V_TEACHER_F16=$f python3 bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip 2>/dev/null | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
    echo "teacher_f16=$f $v" | tee -a gpurun_out/r06_f16_ab.txt
  done
done
