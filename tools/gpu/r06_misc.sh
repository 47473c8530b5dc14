#!/bin/bash
# round 6: whole-step A/B of run-time switches (alternating default-bench runs):  bash tools/gpu/r06_misc.sh "<A2V_DEBUG_SET=..>" ...
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
bash tools/gpu/ab_env.sh 3 "-" "$@" > /dev/null 2>&1
cat gpurun_out/ab_env.txt
