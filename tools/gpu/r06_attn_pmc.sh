#!/bin/bash
# round 6: SQ counters of the student attention kernels, round-5 (39=0) and round-6 (39=1) forms, at the bench's shape
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
out=$root/gpurun_out/r06_attn_pmc.txt; : > $out
for s in ${ARMS:-"39=0" "39=1"}; do
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_FLAT"; do
    echo "== [$s] $set" >> $out
    bash tools/pmc_multi.sh attn_ $set -- $root/tools/attn_bench.py --which student --student-rows 768 --student-s 152 --iters 3 --debug-set $s >> $out 2>&1
  done
done
cat $out
