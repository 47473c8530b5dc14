#!/bin/bash
# round 6: correctness of the five-wave whole-sequence attention kernels, then old / new at the bench's shape (768 rows x 152 kept frames)
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out
python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" > gpurun_out/r06_attn_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06_attn_tests.txt
tail -5 gpurun_out/r06_attn_tests.txt
: > gpurun_out/r06_attn_bench.txt
for r in 1 2 3; do
  for arm in 0 1; do
    for S in 152 140; do
      echo "seq2=$arm S=$S $(python3 tools/attn_bench.py --which student --student-rows 768 --student-s $S --iters 20 --debug-set 39=$arm | tail -1)" | tee -a gpurun_out/r06_attn_bench.txt
    done
  done
done
echo "nodrop" >> gpurun_out/r06_attn_bench.txt
for arm in 0 1; do echo "seq2=$arm S=152 nodrop $(python3 tools/attn_bench.py --which student --student-rows 768 --student-s 152 --iters 20 --no-drop --debug-set 39=$arm | tail -1)" | tee -a gpurun_out/r06_attn_bench.txt; done
echo "hash-only backward (A2V_DROPMASK=0)" >> gpurun_out/r06_attn_bench.txt
A2V_DROPMASK=0 python3 tools/attn_bench.py --which student --student-rows 768 --student-s 152 --iters 20 | tail -1 | tee -a gpurun_out/r06_attn_bench.txt
