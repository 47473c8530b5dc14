set -x
mkdir -p gpurun_out
A2V_BENCH_STEP_TIMES=1 python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip > gpurun_out/b13a.json 2> gpurun_out/b13a.err; grep "step " gpurun_out/b13a.err | head -20; tail -c 150 gpurun_out/b13a.json
python -m pytest tests/test_pretrain_gpu.py tests/test_masks_gpu.py tests/test_dropin_gpu.py -x -q -k "not full_size and not long_sequence and not config5 and not base_star" 2>&1 | tail -4
A2V_BENCH_STEP_TIMES=1 python bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 8 > gpurun_out/b13b.json 2> gpurun_out/b13b.err; grep "step " gpurun_out/b13b.err | head -20; tail -c 150 gpurun_out/b13b.json
