set -x
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b8a.json 2>&1; tail -c 300 gpurun_out/b8a.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b8b.json 2>&1; tail -c 300 gpurun_out/b8b.json
A2V_PARITY_REPORT=gpurun_out/parity_r03g.jsonl timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python bench.py --steps 3 --warmup 1 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b8c.json 2>&1; tail -c 300 gpurun_out/b8c.json
bash tools/gpu/run_profiles.sh r03 2>&1 | grep -v "simple_timer" | tail -30
