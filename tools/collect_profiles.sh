#!/bin/bash
# Collects the per-round evidence on a GPU box: kernel-trace stats of a bench run, the three PMC passes (FETCH_SIZE,
# WRITE_SIZE, MFMA-busy) each in its own run, and writes raw csv under gpurun_out/<tag>_<name>_* (summaries are made
# afterwards by tools/pmc_traffic.py / tools/pmc_mfma.py and copied into profiles/).
#   usage: tools/collect_profiles.sh <tag> [name] [bench args for the stats run] [bench args for the PMC runs] [script]
#   e.g.   tools/collect_profiles.sh r03 large
#          tools/collect_profiles.sh r03 base "--model base --global-batch 64 --steps 2 --warmup 1" "--model base --global-batch 64 --steps 1 --warmup 0"
#          tools/collect_profiles.sh r03 finetune "--steps 2 --warmup 1 --accum 2" "" tools/finetune_bench.py
tag=${1:-rXX}
name=${2:-large}
sargs=${3:---steps 2 --warmup 1 --global-batch 64}
pargs=${4---steps 1 --warmup 0 --global-batch 64}
script=${5:-bench.py}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
export A2V_TEACHER_STREAM=0        # teacher on the main stream: kernel durations are not inflated by overlap
extra=""
[ "$script" = "bench.py" ] && extra="--cpu-baseline skip --no-roofline --parity-leg skip"
cd /tmp && export TMPDIR=/tmp
run() {   # pass name, then rocprofv3 options
  pn=$1; shift
  rm -rf /tmp/cp_$pn
  timeout 600 rocprofv3 "$@" --output-format csv -d /tmp/cp_$pn -- python3 $root/$script $extra $BARGS > /tmp/cp_$pn.log 2>&1 < /dev/null
  tail -1 /tmp/cp_$pn.log | cut -c1-160
}
BARGS="$sargs" run stats --kernel-trace --stats
f=$(find /tmp/cp_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_${name}_kernel_stats.csv
if [ -n "$pargs" ]; then
  BARGS="$pargs" run fetch --kernel-trace --pmc FETCH_SIZE
  f=$(find /tmp/cp_fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_${name}_fetch.csv
  BARGS="$pargs" run write --kernel-trace --pmc WRITE_SIZE
  f=$(find /tmp/cp_write -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_${name}_write.csv
  BARGS="$pargs" run mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  f=$(find /tmp/cp_mfma -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_${name}_mfma.csv
fi
ls -la $out/${tag}_${name}_* 2>/dev/null
