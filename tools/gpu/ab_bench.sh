#!/bin/bash
# Same-box A/B of two builds on the default bench: tools/build_prev.sh <rev> first (animal2vec_amd/liba2v_hip_prev.so), then
#   bash tools/gpu/ab_bench.sh [rounds] [bench args...]      -> gpurun_out/ab_bench.txt (alternating runs, value per run)
rounds=${1:-3}; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ab_bench.txt
cd $root
: > $out
for i in $(seq 1 $rounds); do
  for arm in prev new; do
    if [ $arm = prev ]; then export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_prev.so; else unset A2V_LIB_PATH; fi
    v=$(python3 bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip "$@" 2>/dev/null | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['kept_frames_mean'])")
    echo "$arm $v" | tee -a $out
  done
done
