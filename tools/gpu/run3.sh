set -x
mkdir -p gpurun_out
A2V_PARITY_REPORT=gpurun_out/parity_r03b.jsonl timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 64 > gpurun_out/b3_sampled64.json 2>&1; tail -c 400 gpurun_out/b3_sampled64.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 64 --masks precomputed --keep 140 > gpurun_out/b3_pre64_k140.json 2>&1; tail -c 400 gpurun_out/b3_pre64_k140.json
