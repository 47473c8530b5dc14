#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; cd $root
timeout 600 python3 -m pytest tests/test_ops_gpu.py -k "conv" -x -q 2>&1 | tail -3
(A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_prev.so timeout 300 python3 tools/dec_bench.py; timeout 300 python3 tools/dec_bench.py) > $out/r05_dec_bench.txt 2>&1; cat $out/r05_dec_bench.txt
bash tools/gpu/ab_bench.sh 2
