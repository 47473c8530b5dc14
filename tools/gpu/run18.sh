set -x
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "x3" 2>&1 | tail -4
A2V_PARITY_REPORT=gpurun_out/parity_r03i.jsonl python -m pytest tests/test_pretrain_gpu.py -x -q -k "large_full_size" 2>&1 | tail -4
python tools/x3_bench.py 2>&1 | tail -1
python tools/x3_bench.py --f32 2>&1 | tail -1
