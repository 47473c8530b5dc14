#!/bin/bash
# round 6: output-store policy of the NT GEMM in the whole step: st0 (plain stores) against the product (nt sc1): alternating default
# bench runs, then the kernel stats of each.   -> gpurun_out/ab_env.txt, gpurun_out/r06st{0,4}_large_kernel_stats.csv
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
bash tools/gpu/ab_env.sh ${1:-3} "A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_st0.so" "-" > /dev/null 2>&1
cat gpurun_out/ab_env.txt
A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_st0.so bash tools/collect_profiles.sh r06st0 large "--steps 2 --warmup 1 --global-batch 64" "" | tail -1
cd $root; bash tools/collect_profiles.sh r06st4 large "--steps 2 --warmup 1 --global-batch 64" "" | tail -1
