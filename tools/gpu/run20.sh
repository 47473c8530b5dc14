set -x
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "x3" 2>&1 | tail -3
python -m pytest tests/test_pretrain_gpu.py -x -q -k "large_full_size_vs_oracle and x3 or hd64_vs" 2>&1 | tail -3
python tools/x3_bench.py 2>&1 | tail -1
python tools/x3_bench.py --f32 2>&1 | tail -1
