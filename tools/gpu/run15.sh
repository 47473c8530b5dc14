set -x
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "tail_round or linear" 2>&1 | tail -4
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b15a.json 2>&1; tail -c 150 gpurun_out/b15a.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 --debug-set 22=0 > gpurun_out/b15b.json 2>&1; tail -c 150 gpurun_out/b15b.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b15c.json 2>&1; tail -c 150 gpurun_out/b15c.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 --debug-set 22=0 > gpurun_out/b15d.json 2>&1; tail -c 150 gpurun_out/b15d.json
