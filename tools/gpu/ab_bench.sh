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
    # (a library older than ABI 11 cannot take the 8-bit derivative tensor: A2V_DACT8=0 for that arm -- A2V_PREV_DACT8=1 if it can)
    if [ $arm = prev ]; then export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_prev.so; export A2V_DACT8=${A2V_PREV_DACT8:-0}; else unset A2V_LIB_PATH; unset A2V_DACT8; fi
    v=$(python3 bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip "$@" 2>$out.err | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['kept_frames_mean'])")
    [ -z "$v" ] && tail -3 $out.err
    echo "$arm $v" | tee -a $out
  done
done
