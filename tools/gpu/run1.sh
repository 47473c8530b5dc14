set -x
python -m pytest tests/test_masks_gpu.py -x -q 2>&1 | tail -15
python -m pytest tests/test_pretrain_gpu.py -x -q -k "resume or training_loop or dropin or tiny_fp32" 2>&1 | tail -5
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 64 > gpurun_out/b_sampled64.json 2>gpurun_out/b_sampled64.err; tail -c 1500 gpurun_out/b_sampled64.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 64 --masks precomputed --keep 152 > gpurun_out/b_pre64.json 2>gpurun_out/b_pre64.err; tail -c 600 gpurun_out/b_pre64.json
python bench.py --steps 4 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 64 > gpurun_out/b_sampled64b.json 2>&1; tail -c 300 gpurun_out/b_sampled64b.json
python bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 8 > gpurun_out/b_sampled8.json 2>&1; tail -c 600 gpurun_out/b_sampled8.json
python bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 8 --masks precomputed --keep 144 > gpurun_out/b_pre8.json 2>&1; tail -c 300 gpurun_out/b_pre8.json
python bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-roofline --global-batch 16 > gpurun_out/b_sampled16.json 2>&1; tail -c 300 gpurun_out/b_sampled16.json
