set -x
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "instnorm" 2>&1 | tail -3
python -m pytest tests/test_pretrain_gpu.py -x -q -k "tiny_fp32 or hd64_vs or base_star_pretrain" 2>&1 | tail -3
python tools/elem_bench.py 2>&1 | grep -i "instnorm\|target" | head
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --parity-leg skip --global-batch 64 > gpurun_out/b21.json 2>&1; tail -c 150 gpurun_out/b21.json
