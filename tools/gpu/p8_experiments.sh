#!/bin/bash
# Round-4 experiments on the NT GEMM kernels, one section per call (one GPU box each).  The diagnostic libraries are built first, here
# in the repo (they travel with the snapshot):  python -m animal2vec_amd.build --variant <tag> <defines>   (list per section below).
#   bash tools/gpu/p8_experiments.sh <section>      raw log -> gpurun_out/r04_<section>.txt; summaries by hand -> profiles/r04_*.txt
#   sections:
#     probe    product timing, -DP8_L2STORE=1 (l2s1), start skew (-DP8_SKEW: skew), per-tile stamps (-DP8_STAMP -DP8_STAMP2: probe)
#     stores   -DP8_L2STORE=1|2|3 (l2s1 l2s2 l2s3): what the output stores cost, by pattern and by destination
#     ab       previous build (tools/build_prev.sh <rev>: prev) against the current one + the stamps of the current one
#     kloop    timing builds -DP8_EXP=1|2 (p8e1 p8e2), cache-policy bits -DP8_DMA_AUX=n -DW2_DMA_AUX=n (aux1 aux2 aux4 aux6), stamps against M
#     w2       the 256 x 128 two-per-CU kernel (a2v_debug_set(25, 2)) and its timing builds -DW2_EXP=1|2|4|7 (w2e1 w2e2 w2e4 w2e7)
sec=${1:-probe}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_$sec.txt
cd $root
export PYTHONUNBUFFERED=1
: > $out
lib() { if [ -z "$1" ]; then unset A2V_LIB_PATH; else export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_$1.so; fi; }
probe() { timeout 300 python3 tools/p8_probe.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a $out; }
case $sec in
  probe)
    lib ""; probe
    lib l2s1; probe --only qkv; probe --only fc1
    lib skew; probe --skews 0,1,2,4,6
    lib probe; probe ;;
  stores)
    for l in "" l2s1 l2s2 l2s3 "" l2s1 l2s2 l2s3; do lib "$l"; echo "## lib=$l" | tee -a $out; probe --only qkv; probe --only fc1; done ;;
  ab)
    timeout 600 python3 -m pytest tests/test_ops_gpu.py -x -q -k "p8 or linear or gemm or dact" 2>&1 | tail -3 | tee -a $out
    for l in prev "" prev ""; do lib "$l"; echo "## lib=$l" | tee -a $out; probe; done
    lib probe; probe
    lib ""; bash tools/gpu/ab_bench.sh 3 > /dev/null 2>&1; cat gpurun_out/ab_bench.txt | tee -a $out ;;
  kloop)
    for l in "" p8e1 p8e2; do lib "$l"; echo "## lib=$l" | tee -a $out; probe --only qkv; probe --only "fc2   bias"; done
    for l in "" aux1 aux2 aux4 aux6; do for s in "" 25=2; do
      lib "$l"; if [ -z "$s" ]; then unset A2V_DEBUG_SET; else export A2V_DEBUG_SET=$s; fi
      echo "## lib=$l set=$s" | tee -a $out; probe --only qkv; probe --only "fc2   bias"; done; done
    unset A2V_DEBUG_SET; lib probe
    for M in 8192 16384 32768 116736; do echo "## M=$M" | tee -a $out; probe --M $M --only qkv; probe --M $M --only "fc2   bias"; done ;;
  w2)
    A2V_DEBUG_SET=25=2 timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "p8 or linear or gemm or dact" 2>&1 | tail -3 | tee -a $out
    for s in "" 25=2 "" 25=2; do if [ -z "$s" ]; then unset A2V_DEBUG_SET; else export A2V_DEBUG_SET=$s; fi; echo "## A2V_DEBUG_SET=$s" | tee -a $out; lib ""; probe; done
    export A2V_DEBUG_SET=25=2
    for l in w2e1 w2e2 w2e4 w2e7; do lib "$l"; echo "## lib=$l" | tee -a $out; probe --only qkv; probe --only "fc2   bias"; done ;;
  *) echo "unknown section $sec"; exit 1 ;;
esac
