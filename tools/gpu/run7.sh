set -x
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "x3 or split" 2>&1 | tail -8
A2V_PARITY_REPORT=gpurun_out/parity_r03f.jsonl python -m pytest tests/test_pretrain_gpu.py -x -q -k "hd64_vs or large_full_size" 2>&1 | tail -8
python bench.py --steps 3 --warmup 1 --cpu-baseline skip --no-roofline --global-batch 64 > gpurun_out/b7.json 2>&1; tail -c 900 gpurun_out/b7.json
