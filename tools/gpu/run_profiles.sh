#!/bin/bash
# per-round evidence on one box (bash tools/gpu/run_profiles.sh <tag> [final]): kernel stats of the headline config and of the 8- / 16-clip steps one rank runs at 8 GPUs, the
# three PMC passes of the headline config, the bf16x3 mode's kernel stats, the step times at 128 / 16 / 8 clips and the default
# bench line.  raw csv / logs -> gpurun_out/r05_*
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd $root
bash tools/collect_profiles.sh $tag large "--steps 2 --warmup 1 --global-batch 64" "--steps 1 --warmup 0 --global-batch 64"
bash tools/collect_profiles.sh $tag large_mb8 "--steps 6 --warmup 2 --global-batch 8" ""
bash tools/collect_profiles.sh $tag large_mb16 "--steps 6 --warmup 2 --global-batch 16" ""
bash tools/collect_profiles.sh $tag large_x3 "" "" tools/x3_bench.py
cd $root
for gb in 16 8; do
  timeout 300 python3 bench.py --global-batch $gb --steps 10 --warmup 3 --cpu-baseline skip --no-roofline --parity-leg skip 2>/dev/null | grep '^{' > $out/${tag}_bench_gb$gb.json
done
timeout 600 python3 bench.py > $out/${tag}_bench_default.log 2>&1
grep '^{' $out/${tag}_bench_default.log > $out/${tag}_bench_default.json

# (the end-of-round pass, on the final build: GPU suite with the parity report + the bf16x3 mode's stats again)
if [ "$2" = "final" ]; then
  rm -f $out/${tag}_parity_report.jsonl
  A2V_PARITY_REPORT=$out/${tag}_parity_report.jsonl timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 > $out/${tag}_final_gputest.log
  tail -2 $out/${tag}_final_gputest.log
fi
