set -x
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "strided or conv" 2>&1 | tail -8
A2V_PARITY_REPORT=gpurun_out/parity_r03c.jsonl python -m pytest tests/test_pretrain_gpu.py -q -k "large_full_size or long_sequence or base_star_pretrain or hd64_vs or config5 or tiny_fp32 or training_loop" 2>&1 | tail -8
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 64 > gpurun_out/b4_strided.json 2>&1; tail -c 300 gpurun_out/b4_strided.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 64 --debug-set 21=0 > gpurun_out/b4_nostrided.json 2>&1; tail -c 300 gpurun_out/b4_nostrided.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 64 > gpurun_out/b4_strided2.json 2>&1; tail -c 300 gpurun_out/b4_strided2.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 64 --debug-set 21=0 > gpurun_out/b4_nostrided2.json 2>&1; tail -c 300 gpurun_out/b4_nostrided2.json
