set -x
mkdir -p gpurun_out
A2V_BENCH_FORCE_DIST=1 A2V_DDP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 16 > gpurun_out/b17_dist16.json 2>&1; tail -c 200 gpurun_out/b17_dist16.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 16 > gpurun_out/b17_16.json 2>&1; tail -c 200 gpurun_out/b17_16.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 --masks precomputed --keep 140 > gpurun_out/b17_pre140.json 2>&1; tail -c 200 gpurun_out/b17_pre140.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b17_64.json 2>&1; tail -c 200 gpurun_out/b17_64.json
bash tools/collect_profiles.sh r03 large_mb16 "--steps 6 --warmup 2 --global-batch 16" "" 2>&1 | tail -2
