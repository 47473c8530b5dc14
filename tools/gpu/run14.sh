set -x
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "attention or alibi" 2>&1 | tail -4
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b14a.json 2>&1; tail -c 150 gpurun_out/b14a.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 --debug-set 15=0 > gpurun_out/b14b.json 2>&1; tail -c 150 gpurun_out/b14b.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b14c.json 2>&1; tail -c 150 gpurun_out/b14c.json
python tools/attn_bench.py --iters 3 --which student 2>&1 | tail -6
