set -x
mkdir -p gpurun_out
A2V_BENCH_STEP_TIMES=1 python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip > gpurun_out/b12a.json 2> gpurun_out/b12a.err; grep "step \|arn" gpurun_out/b12a.err | head -20; tail -c 150 gpurun_out/b12a.json
A2V_BENCH_STEP_TIMES=1 PYTORCH_HIP_ALLOC_CONF=expandable_segments:True python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip > gpurun_out/b12b.json 2> gpurun_out/b12b.err; grep "step \|arn\|xpand" gpurun_out/b12b.err | head -20; tail -c 150 gpurun_out/b12b.json
A2V_BENCH_STEP_TIMES=1 PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip > gpurun_out/b12c.json 2> gpurun_out/b12c.err; grep "step \|arn\|xpand" gpurun_out/b12c.err | head -20; tail -c 150 gpurun_out/b12c.json
