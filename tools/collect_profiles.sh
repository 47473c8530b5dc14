#!/bin/bash
# Collects the per-round evidence on a GPU box: kernel-trace stats of the default bench, the three PMC passes (FETCH_SIZE,
# WRITE_SIZE, MFMA-busy) each in its own run, and writes raw csv under gpurun_out/<tag>_* (summaries are made afterwards by
# tools/pmc_traffic.py / tools/pmc_mfma.py and copied into profiles/).   usage: tools/collect_profiles.sh <tag>
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
export A2V_TEACHER_STREAM=0        # teacher on the main stream: kernel durations are not inflated by overlap
cd /tmp && export TMPDIR=/tmp
run() {   # name, then rocprofv3 options
  name=$1; shift
  rm -rf /tmp/cp_$name
  timeout 400 rocprofv3 "$@" --output-format csv -d /tmp/cp_$name -- python3 $root/bench.py --cpu-baseline skip --no-roofline $BARGS > /tmp/cp_$name.log 2>&1 < /dev/null
  tail -1 /tmp/cp_$name.log | cut -c1-120
}
BARGS="--steps 2 --warmup 1" run stats --kernel-trace --stats
f=$(find /tmp/cp_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats.csv
BARGS="--steps 1 --warmup 0 --global-batch 64" run fetch --kernel-trace --pmc FETCH_SIZE
f=$(find /tmp/cp_fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_fetch.csv
BARGS="--steps 1 --warmup 0 --global-batch 64" run write --kernel-trace --pmc WRITE_SIZE
f=$(find /tmp/cp_write -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_write.csv
BARGS="--steps 1 --warmup 0 --global-batch 64" run mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
f=$(find /tmp/cp_mfma -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_mfma.csv
ls -la $out/${tag}_* 2>/dev/null
