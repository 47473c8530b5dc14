set -x
mkdir -p gpurun_out
A2V_BENCH_STEP_TIMES=1 python bench.py --steps 3 --warmup 1 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b9a.json 2> gpurun_out/b9a.err; grep "step " gpurun_out/b9a.err; tail -c 200 gpurun_out/b9a.json
A2V_BENCH_STEP_TIMES=1 python bench.py --steps 3 --warmup 1 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b9b.json 2> gpurun_out/b9b.err; grep "step " gpurun_out/b9b.err; tail -c 200 gpurun_out/b9b.json
python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | tail -4
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b9c.json 2>&1; tail -c 200 gpurun_out/b9c.json
python bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 8 > gpurun_out/b9d.json 2>&1; tail -c 200 gpurun_out/b9d.json
