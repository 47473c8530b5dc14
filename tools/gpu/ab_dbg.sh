#!/bin/bash
# Same-box A/B of a2v_debug_set switches on the default bench:  bash tools/gpu/ab_dbg.sh <rounds> "<set A>" "<set B>" ... [-- bench args]
# (a set is a comma list of key=value, "-" = none)  -> gpurun_out/ab_dbg.txt
rounds=$1; shift
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
[ "$1" = "--" ] && shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ab_dbg.txt
cd $root
: > $out
for i in $(seq 1 $rounds); do
  for s in "${sets[@]}"; do
    if [ "$s" = "-" ]; then unset A2V_DEBUG_SET; else export A2V_DEBUG_SET="$s"; fi
    v=$(python3 bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip "$@" 2>/dev/null | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['kept_frames_mean'])")
    echo "[$s] $v" | tee -a $out
  done
done
