# round-3 evidence: kernel stats (+ PMC passes for the headline config) of every BASELINE config, raw csv -> gpurun_out/r03_*
set -x
tag=${1:-r03}
bash tools/collect_profiles.sh $tag large "--steps 2 --warmup 1 --global-batch 64" "--steps 1 --warmup 0 --global-batch 64"
bash tools/collect_profiles.sh $tag large_mb8 "--steps 6 --warmup 2 --global-batch 8" ""
bash tools/collect_profiles.sh $tag base "--model base --steps 2 --warmup 1 --global-batch 64" "--model base --steps 1 --warmup 0 --global-batch 64"
bash tools/collect_profiles.sh $tag large44k "--model large44k --samples 441000 --global-batch 4 --micro-batch 2 --steps 2 --warmup 1" "--model large44k --samples 441000 --global-batch 2 --micro-batch 2 --steps 1 --warmup 0"
bash tools/collect_profiles.sh $tag finetune "--steps 2 --warmup 1 --accum 2" "" tools/finetune_bench.py
bash tools/collect_profiles.sh $tag large_fp32 "--dtype fp32 --steps 1 --warmup 1 --global-batch 8" ""
