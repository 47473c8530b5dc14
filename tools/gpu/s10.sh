#!/bin/bash
# round 5: SQ counters of the long-sequence attention forwards (tools/pmc_pass.sh, one pass per counter set)
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
out=$root/gpurun_out/s10_attn_pmc.txt; : > $out
for s in "36=0" "36=1 --debug-set 18=0"; do
  k=attn_fwd_bf16_kernel; [ "$s" != "36=0" ] && k=attn_fwd32_bf16_kernel
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
    echo "== [$s] $set" >> $out
    bash tools/pmc_pass.sh $k $set -- $root/tools/attn_bench.py --which teacher --iters 3 --debug-set $s >> $out 2>&1
  done
done
cat $out
