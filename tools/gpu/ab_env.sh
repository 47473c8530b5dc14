#!/bin/bash
# Same-box A/B of environment switches on the default bench:  bash tools/gpu/ab_env.sh <rounds> "<VAR=val ...>" "<VAR=val ...>" ... [-- bench args]
# ("-" = nothing set)  -> gpurun_out/ab_env.txt
rounds=$1; shift
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
[ "$1" = "--" ] && shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ab_env.txt
cd $root
: > $out
for i in $(seq 1 $rounds); do
  for s in "${sets[@]}"; do
    if [ "$s" = "-" ]; then e=(); else read -r -a e <<< "$s"; fi
    v=$(env "${e[@]}" python3 bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip "$@" 2>/dev/null | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['kept_frames_mean'])")
    echo "[$s] $v" | tee -a $out
  done
done
